"""CPU tests that pin the SimCLR oracle (oracle/simclr.py): (1) against golden vectors produced
by running the reference's own SimCLR sources (tests/golden/make_golden_simclr.py), (2) live
against those sources when /root/reference is present, (3) fp64 / closed-form known answers for
the NT-Xent + CO2 head, LARS and the warm-up schedule."""
import math
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import simclr as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')
SOLVER = dict(T=0.1, lr=4.0, warmup_steps=2, t_max=1000)      # make_golden_simclr.SOLVER
WATCH = ['0.conv1.weight', '0.layer1.0.conv2.weight', '0.layer2.0.downsample.0.weight',
         '0.layer4.2.conv3.weight', '0.layer3.5.bn2.weight', '0.bn1.bias',
         '1.mlp.0.weight', '1.mlp.3.bias', '1.mlp.6.weight', '1.mlp.7.weight']
WATCH_STATS = ['0.bn1._mean', '0.bn1._variance', '1.mlp.7._mean', '1.mlp.7._variance']


def test_oracle_matches_golden_small():
    z = np.load(os.path.join(GOLDEN, 'simclr_r50_small.npz'))
    N, hw, steps = [int(v) for v in z['meta']]
    o = S.SimCLROracle(seed=0, **SOLVER)
    gen = torch.Generator().manual_seed(4321)
    for s in range(2):                      # step 2 is past the first LARS update: chaotic
        xq = torch.randn(N, 3, hw, hw, generator=gen)
        xk = torch.randn(N, 3, hw, hw, generator=gen)
        lr = o.lr()
        out = o.train_step(xq, xk)
        pre = 's%d_' % s
        assert abs(lr - float(z[pre + 'lr'])) < 1e-12
        assert abs(float(out['loss']) - float(z[pre + 'loss'])) < 2e-5
        assert float(out['acc1']) == float(z[pre + 'acc1'])
        np.testing.assert_allclose(out['q'][:, :8].numpy(), z[pre + 'q_head'], atol=2e-6)
        np.testing.assert_allclose(out['mats']['ab'][:, :8].numpy(), z[pre + 'ab_head'], atol=2e-5)
        for n in WATCH:
            g = out['grads'][n].double().norm().item()
            assert abs(g - float(z[pre + 'gradnorm/' + n])) <= 2e-4 * max(g, 1e-6), n
            assert abs(o.st[n].double().norm().item() - float(z[pre + 'pnorm/' + n])) < 1e-4, n
        for n in WATCH_STATS:
            np.testing.assert_allclose(o.st[n][:8].numpy(), z[pre + 'stat/' + n], atol=1e-5)


@pytest.mark.skipif(not os.path.isdir('/root/reference/passl_v110'),
                    reason='reference tree not present (GPU box)')
def test_oracle_matches_reference_sources_live():
    """Executes the reference's simclr.py / simclr_contrastive_head.py / resnetsimclr.py under the
    paddle shim in a subprocess and compares loss, accuracy and every gradient."""
    code = r'''
import torch
from oracle import ref_runner
from oracle.simclr import SimCLROracle
o = SimCLROracle(seed=1, lr=2.0, warmup_steps=2, t_max=100)
m = ref_runner.build_reference_simclr()
ref_runner.load_simclr_state(m, o)
m.train()
g = torch.Generator().manual_seed(5)
xq = torch.randn(6, 3, 48, 40, generator=g); xk = torch.randn(6, 3, 48, 40, generator=g)
res = m(xq, xk, mode='train')
res['loss'].backward()
out = o.train_step(xq, xk)
assert abs(float(res['loss'].detach()) - float(out['loss'])) < 1e-6
assert float(res['acc1']) == float(out['acc1'])
ps = dict(m.encoder.named_parameters())
for n, gr in out['grads'].items():
    assert (ps[n].grad - gr).abs().max().item() <= 1e-6 * max(gr.abs().max().item(), 1.0), n
print('LIVE-OK')
'''
    r = subprocess.run([sys.executable, '-c', code], cwd=ROOT, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and 'LIVE-OK' in r.stdout, r.stderr[-3000:]


def test_head_fp32_vs_fp64_and_invariants():
    gen = torch.Generator().manual_seed(3)
    for B in (5, 16, 33):
        a = torch.nn.functional.normalize(torch.randn(B, 128, generator=gen), dim=1)
        b = torch.nn.functional.normalize(torch.randn(B, 128, generator=gen), dim=1)
        loss, acc1, mats = S.simclr_head(a, b, 0.1)
        l64, a64 = S.simclr_head_f64(a.numpy(), b.numpy(), 0.1)
        assert abs(float(loss) - l64) < 2e-5 and abs(float(acc1) - a64) < 1e-6
        # identical views: the positive is the arg-max of every row (acc 1) and CO2 vanishes
        loss_same, acc_same, _ = S.simclr_head(a, a.clone(), 0.1)
        ce_only = S.simclr_head(a, a.clone(), 0.1, co2_weight=0.0)[0]
        assert float(acc_same) == 1.0 and abs(float(loss_same) - float(ce_only)) < 1e-5
        # symmetric in (a, b)
        assert abs(float(S.simclr_head(b, a, 0.1)[0]) - float(loss)) < 1e-5
        assert float(mats['aa'].diagonal().max()) < -1e8           # LARGE_NUM self-mask


def test_head_gradient_closed_form():
    """The kernel's backward uses these coefficient matrices (kl_div target carries no gradient)."""
    gen = torch.Generator().manual_seed(0)
    B, T = 12, 0.1
    a = torch.nn.functional.normalize(torch.randn(B, 16, generator=gen, dtype=torch.float64), dim=1)
    b = torch.nn.functional.normalize(torch.randn(B, 16, generator=gen, dtype=torch.float64), dim=1)
    a.requires_grad_(True); b.requires_grad_(True)
    S.simclr_head(a, b, T)[0].backward()
    with torch.no_grad():
        A, Bm = a.detach(), b.detach()
        eye = torch.eye(B, dtype=torch.bool)
        ninf = float('-inf')
        aa, ab, ba, bb = A @ A.t() / T, A @ Bm.t() / T, Bm @ A.t() / T, Bm @ Bm.t() / T
        aam, bbm = aa.masked_fill(eye, ninf), bb.masked_fill(eye, ninf)
        abm, bam = ab.masked_fill(eye, ninf), ba.masked_fill(eye, ninf)
        lce_a = torch.logsumexp(torch.cat([ab, aam], 1), 1)[:, None]
        lce_b = torch.logsumexp(torch.cat([ba, bbm], 1), 1)[:, None]
        lx = torch.logsumexp(torch.cat([aam, abm], 1), 1)[:, None]
        ly = torch.logsumexp(torch.cat([bam, bbm], 1), 1)[:, None]
        pa_aa, pa_ab = torch.exp(aam - lx), torch.exp(abm - lx)
        pb_ba, pb_bb = torch.exp(bam - ly), torch.exp(bbm - ly)
        d = eye.double()
        g_aa = (torch.exp(aam - lce_a) + 3 * (pa_aa - pb_ba)) / B
        g_ab = (torch.exp(ab - lce_a) - d + 3 * (pa_ab - pb_bb)) / B
        g_ba = (torch.exp(ba - lce_b) - d + 3 * (pb_ba - pa_aa)) / B
        g_bb = (torch.exp(bbm - lce_b) + 3 * (pb_bb - pa_ab)) / B
        da = (g_aa @ A + g_ab @ Bm + g_aa.t() @ A + g_ba.t() @ Bm) / T
        db = (g_ba @ A + g_bb @ Bm + g_ab.t() @ A + g_bb.t() @ Bm) / T
    assert (da - a.grad).abs().max() < 1e-12 and (db - b.grad).abs().max() < 1e-12


def test_lars_and_schedule_known_answers():
    # schedule: linear warm-up from 0, then the cosine counted from the end of the warm-up
    assert S.simclr_lr(0, 64.0, 10, 100) == 0.0
    assert abs(S.simclr_lr(5, 64.0, 10, 100) - 32.0) < 1e-12
    assert abs(S.simclr_lr(10, 64.0, 10, 100) - 64.0) < 1e-12
    assert abs(S.simclr_lr(60, 64.0, 10, 100) - 32.0) < 1e-9
    # trainer.py:161-163 / builder.py:54-66 numbers of configs/simclr/simclr_r50_IM.yaml
    bs = 512 * 8
    warm = int(round(10 * 1281167 // bs))
    total = 1281167 * 100 // bs + 1
    assert (warm, total - warm, 1.0 * math.sqrt(bs)) == (3127, 28152, 64.0)
    # LARS: one step on a single tensor, then momentum
    o = S.SimCLROracle(seed=0, lr=2.0, warmup_steps=1, t_max=10)
    o.step_count = 1
    keys = list(o.st.keys())
    n = '0.conv1.weight'
    p0 = o.st[n].clone()
    g = torch.ones_like(p0) * 0.01
    o.apply_lars({n: g})
    pn, gn = p0.double().norm().item(), g.double().norm().item()
    local = 2.0 * 0.001 * pn / (gn + 1e-4 * pn)
    v = local * (g + 1e-4 * p0)
    assert (o.st[n] - (p0 - v)).abs().max() < 1e-7
    assert len(o.excluded) == 0 and keys[0] == n      # yaml's exclude list matches no Paddle name
    # zero gradient -> local_lr = lr (the op's fall-back branch), update = lr*wd*p
    p1 = o.st[n].clone()
    o.apply_lars({n: torch.zeros_like(p1)})
    lr2 = S.simclr_lr(2, 2.0, 1, 10)
    assert (o.st[n] - (p1 - (0.9 * v + lr2 * 1e-4 * p1))).abs().max() < 1e-6


@pytest.mark.skipif(not os.path.isdir('/root/reference/passl_v110'),
                    reason='reference tree not present (GPU box)')
def test_reference_cannot_build_its_own_r18_cifar10_yaml_but_builds_the_registered_class_over_the_same_file():
    """configs/simclr/simclr_r18_cifar10.yaml names ``backbone: ResNetCifar`` — in the reference that is an import alias
    (backbones/__init__.py:15), never registered, and its lr block gives ``CosineWarmup`` keys its __init__ does not
    take.  Pinned by running the reference's own builders on the YAML (paddle shim, subprocess); the registered class
    over the same file, ``ResNetsimclr(depth=18)``, builds the BasicBlock trunk whose state layout the product and
    tests/golden/simclr_r18_*.npz use."""
    code = r'''
import copy, yaml, torch
from oracle import ref_runner
from oracle.simclr import init_encoder_state
ns = ref_runner.load()
cfg = yaml.safe_load(open('/root/reference/configs/simclr/simclr_r18_cifar10.yaml'))
try:
    ns.build_model(copy.deepcopy(cfg['model']))
    raise SystemExit('the reference built ResNetCifar?')
except KeyError as e:
    assert 'ResNetCifar' in str(e), e
src = open('/root/reference/passl_v110/solver/lr_scheduler.py').read()
i = src.index('def __init__', src.index('class CosineWarmup')); sig = src[i:src.index('):', i)]
for key in ('learning_rate_scaling', 'total_images', 'warmup_epochs'):
    assert key in cfg['lr_scheduler'] and key not in sig, key              # kwargs CosineWarmup.__init__ lacks
for need in ('learning_rate', 'warmup_steps'):
    assert need in sig and need not in cfg['lr_scheduler'], need           # ... and required ones the yaml lacks
c2 = copy.deepcopy(cfg['model']); c2['backbone']['name'] = 'ResNetsimclr'
m = ns.build_model(c2)
ost = init_encoder_state(torch.Generator().manual_seed(0), 512, 512, depth=18)
sd = m.encoder.state_dict()
assert list(sd.keys()) == list(ost.keys()) and all(tuple(sd[k].shape) == tuple(ost[k].shape) for k in ost)
assert sum(1 for p in m.parameters() if p.requires_grad) == 12          # frozen_stages: 4 -> the projector only
print('R18-OK')
'''
    r = subprocess.run([sys.executable, '-c', code], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'R18-OK' in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
