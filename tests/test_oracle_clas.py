"""CPU tests that pin the linear-probe oracle (oracle/clas.py): against golden vectors produced by
running the reference's own Classification / ClasHead / frozen ResNet sources
(tests/golden/make_golden_clas.py), live against those sources when /root/reference is present, and
known answers for accuracy / schedule."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import clas as C

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')
LR, MU = 0.002, 0.9


def _against(name, max_steps):
    z = np.load(os.path.join(GOLDEN, name + '.npz'))
    N, size, ncls, steps = [int(v) for v in z['meta']]
    o = C.ClasOracle(num_classes=ncls, seed=0, lr=LR, momentum=MU)
    gen = torch.Generator().manual_seed(909)
    for s in range(min(steps, max_steps)):
        img = torch.randn(N, 3, size, size, generator=gen)
        lab = torch.randint(0, ncls, (N,), generator=gen)
        out = o.train_step(img, lab)
        pre = 's%d_' % s
        assert abs(float(out['loss']) - float(z[pre + 'loss'])) < 2e-5
        assert abs(float(out['acc1']) - float(z[pre + 'acc1'])) < 1e-9 and abs(float(out['acc5']) - float(z[pre + 'acc5'])) < 1e-9
        np.testing.assert_allclose(out['scores'].numpy()[:, :16], z[pre + 'scores'], atol=2e-5)
        for n in ('head.fc_cls.weight', 'head.fc_cls.bias'):
            g = out['grads'][n].double().norm().item()
            assert abs(g - float(z[pre + 'gradnorm/' + n])) <= 1e-4 * max(g, 1e-9), n
            assert abs(o.st[n].double().norm().item() - float(z[pre + 'pnorm/' + n])) < 1e-5, n


def test_oracle_matches_golden_partially_frozen():
    """frozen_stages = 2 (resnet.py:90-106): stem + layer1-2 frozen (parameters AND BatchNorm statistics),
    layer3-4 + head trained with batch statistics — golden from the reference's own code."""
    z = np.load(os.path.join(GOLDEN, 'clas_r50_frozen2.npz'))
    N, size, ncls, steps = [int(v) for v in z['meta']]
    o = C.ClasOracle(num_classes=ncls, seed=0, lr=LR, momentum=MU, frozen_stages=2)
    gen = torch.Generator().manual_seed(909)
    for s in range(steps):
        img = torch.randn(N, 3, size, size, generator=gen)
        lab = torch.randint(0, ncls, (N,), generator=gen)
        out = o.train_step(img, lab)
        pre = 's%d_' % s
        assert abs(float(out['loss']) - float(z[pre + 'loss'])) < (2e-5 if s == 0 else 2e-3)
        for key in z.files:
            if key.startswith(pre + 'gradnorm/'):
                n = key[len(pre + 'gradnorm/'):]
                g = out['grads'][n].double().norm().item()
                assert abs(g - float(z[key])) <= (1e-4 if s == 0 else 2e-2) * max(g, 1e-9), n
            if key.startswith(pre + 'stat/'):
                n = key[len(pre + 'stat/'):]
                np.testing.assert_allclose(o.st[n][:8].double().numpy(), z[key], atol=1e-5 if s == 0 else 1e-3)
    # the frozen stages' statistics never moved, the training ones did
    o0 = C.ClasOracle(num_classes=ncls, seed=0, lr=LR, momentum=MU, frozen_stages=2)
    assert torch.equal(o.st['backbone.layer2.3.bn3._variance'], o0.st['backbone.layer2.3.bn3._variance'])
    assert not torch.equal(o.st['backbone.layer3.0.bn1._mean'], o0.st['backbone.layer3.0.bn1._mean'])
    assert torch.equal(o.st['backbone.layer1.0.conv1.weight'], o0.st['backbone.layer1.0.conv1.weight'])


def test_oracle_matches_golden_small():
    _against('clas_r50_small', 3)


def test_oracle_matches_golden_b16_first_step():
    _against('clas_r50_b16', 1)


@pytest.mark.skipif(not os.path.isdir('/root/reference/passl_v110'),
                    reason='reference tree not present (GPU box)')
def test_oracle_matches_reference_sources_live():
    code = r'''
import torch
from oracle import ref_runner, clas as oc
m = ref_runner.build_reference_clas(num_classes=30)
o = oc.ClasOracle(num_classes=30, seed=7)
ref_runner.load_clas_state(m, o)
m.train()
assert sorted(n for n, p in m.named_parameters() if p.requires_grad) == ['head.fc_cls.bias', 'head.fc_cls.weight']
g = torch.Generator().manual_seed(2)
img = torch.randn(5, 3, 64, 64, generator=g); lab = torch.randint(0, 30, (5,), generator=g)
out = m(img, lab); out['loss'].backward()
r = o.train_step(img, lab)
assert abs(float(out['loss'].detach()) - float(r['loss'])) < 1e-6
assert float(out['acc1']) == float(r['acc1']) and float(out['acc5']) == float(r['acc5'])
ps = dict(m.named_parameters())
for n, gr in r['grads'].items():
    assert (ps[n].grad - gr).abs().max().item() <= 1e-6 * max(gr.abs().max().item(), 1.0), n
assert (m(img, lab, mode='test') - r['scores']).abs().max().item() < 1e-5
# BatchNorm layers really use the running statistics in train() mode (frozen_stages = 4)
o2 = oc.ClasOracle(num_classes=30, seed=7)
r2 = o2.train_step(img[:2], lab[:2])
assert (r2['scores'] - r['scores'][:2]).abs().max().item() < 1e-5      # batch-composition independent
print('LIVE-OK')
'''
    r = subprocess.run([sys.executable, '-c', code], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'LIVE-OK' in r.stdout, r.stderr[-3000:]


def test_accuracy_and_schedule_known_answers():
    s = torch.tensor([[0.1, 0.9, 0.3, 0.2, 0.0, -1.0, 0.5],      # label 1 = top-1
                      [0.1, 0.9, 0.3, 0.2, 0.0, -1.0, 0.5],      # label 4: rank 5 -> outside top-5
                      [0.1, 0.9, 0.3, 0.2, 0.0, -1.0, 0.5],      # label 0: rank 4 -> inside top-5
                      [0.5, 0.5, 0.5, 0.5, 0.5, 0.5, 0.5]])      # all tied, label 5: rank 5 -> outside
    a1, a5 = C.accuracy(s, torch.tensor([1, 4, 0, 5]))
    assert float(a1) == 25.0 and float(a5) == 50.0
    # MultiStepDecay of configs/moco/moco_clas_r50.yaml (epochs)
    assert C.multistep_lr(30.0, 59, [60, 80]) == 30.0
    assert abs(C.multistep_lr(30.0, 60, [60, 80]) - 3.0) < 1e-12 and abs(C.multistep_lr(30.0, 99, [60, 80]) - 0.3) < 1e-12
