"""Generate tests/golden/clas_*.npz by EXECUTING THE REFERENCE's linear-probe sources
(passl_v110/modeling/architectures/clas.py, heads/clas_head.py, backbones/resnet.py with
frozen_stages=4) on torch-CPU through the paddle shim (oracle/ref_runner.py); backward = torch autograd
over the reference's forward graph, momentum-SGD = oracle.clas.ClasOracle's rule (Paddle's optimizer
kernel is not in the reference tree).

    python tests/golden/make_golden_clas.py

Seed-defined inputs (regenerable without /root/reference):
    weights: oracle.clas.ClasOracle(num_classes, seed=0);  per step: img ~ N(0,1), labels ~ U{0..C-1}
    from torch.Generator().manual_seed(909)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import ref_runner                      # noqa: E402
from oracle.clas import ClasOracle                 # noqa: E402

CASES = {
    'clas_r50_small': dict(N=8, size=64, num_classes=48, steps=3),
    # configs/moco/moco_clas_r50.yaml shapes (1000 classes, 224^2), small batch
    'clas_r50_b16': dict(N=16, size=224, num_classes=1000, steps=2),
    # partially frozen trunk (resnet.py:90-106): stem + layer1-2 frozen, layer3-4 + head trained
    'clas_r50_frozen2': dict(N=8, size=64, num_classes=48, steps=2, frozen_stages=2),
}
LR, MU = 0.002, 0.9


WATCH_FROZEN2 = ['backbone.layer3.0.conv1.weight', 'backbone.layer3.0.downsample.0.weight',
                 'backbone.layer3.5.bn2.weight', 'backbone.layer4.2.conv3.weight', 'backbone.layer4.2.bn3.bias',
                 'head.fc_cls.weight', 'head.fc_cls.bias']


def run_case(name, N, size, num_classes, steps, frozen_stages=4):
    torch.manual_seed(0)
    oracle = ClasOracle(num_classes=num_classes, seed=0, lr=LR, momentum=MU, frozen_stages=frozen_stages)
    model = ref_runner.build_reference_clas(num_classes=num_classes, frozen_stages=frozen_stages)
    ref_runner.load_clas_state(model, oracle)
    model.train()
    gen = torch.Generator().manual_seed(909)
    vel = {}
    out = {}
    for s in range(steps):
        img = torch.randn(N, 3, size, size, generator=gen)
        lab = torch.randint(0, num_classes, (N,), generator=gen)
        for p in model.parameters():
            p.grad = None
        res = model(img, lab)
        res['loss'].backward()
        # (with a partially frozen trunk a second forward in train() mode would update the training
        # stages' BatchNorm statistics once more: take the scores from the training forward's inputs only
        # for the fully frozen configs)
        scores = model(img, lab, mode='test') if frozen_stages >= 4 else None
        ps = {n: p for n, p in model.named_parameters() if p.requires_grad}
        if frozen_stages >= 4:
            assert sorted(ps) == ['head.fc_cls.bias', 'head.fc_cls.weight']
        else:
            from oracle.clas import frozen_keys
            assert set(ps) == set(oracle.st) - frozen_keys(oracle.st, frozen_stages), \
                set(ps) ^ (set(oracle.st) - frozen_keys(oracle.st, frozen_stages))
        pre = 's%d_' % s
        out[pre + 'loss'] = np.float64(res['loss'].item())
        out[pre + 'acc1'] = np.float64(float(res['acc1']))
        out[pre + 'acc5'] = np.float64(float(res['acc5']))
        if scores is not None:
            out[pre + 'scores'] = scores.detach().numpy()[:, :16].copy()
        watch = set(ps) if frozen_stages >= 4 else set(WATCH_FROZEN2)
        with torch.no_grad():
            for n, p in ps.items():
                if n in watch:
                    out[pre + 'gradnorm/' + n] = np.float64(p.grad.double().norm().item())
                v = MU * vel.get(n, torch.zeros_like(p)) + p.grad
                vel[n] = v
                p.sub_(LR * v)
                if n in watch:
                    out[pre + 'pnorm/' + n] = np.float64(p.double().norm().item())
            if frozen_stages < 4:      # running statistics: frozen ones unchanged, training ones updated
                sd = model.state_dict()
                for n in ('backbone.layer1.0.bn1._mean', 'backbone.layer2.3.bn3._variance',
                          'backbone.layer3.0.bn1._mean', 'backbone.layer4.2.bn3._variance'):
                    out[pre + 'stat/' + n] = sd[n][:8].numpy().astype(np.float64)
        print(name, 'step', s, 'loss %.6f acc1 %.2f acc5 %.2f' % (out[pre + 'loss'], out[pre + 'acc1'], out[pre + 'acc5']))
    o64 = ClasOracle(num_classes=num_classes, seed=0, lr=LR, momentum=MU, dtype=torch.float64,
                     frozen_stages=frozen_stages)
    gen = torch.Generator().manual_seed(909)
    for s in range(steps):
        img = torch.randn(N, 3, size, size, generator=gen)
        lab = torch.randint(0, num_classes, (N,), generator=gen)
        r = o64.train_step(img.double(), lab)
        pre = 's%d_f64_' % s
        out[pre + 'loss'] = np.float64(float(r['loss']))
        out[pre + 'scores'] = r['scores'].numpy()[:, :16].copy()
        out[pre + 'feat'] = r['feat'].numpy()[:, :16].copy()
        for n, g in r['grads'].items():
            if frozen_stages >= 4 or n in WATCH_FROZEN2:
                out[pre + 'gradnorm/' + n] = np.float64(g.norm().item())
                out[pre + 'pnorm/' + n] = np.float64(o64.st[n].norm().item())
    out['meta'] = np.array([N, size, num_classes, steps], dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)


if __name__ == '__main__':
    assert ref_runner.available(), 'needs /root/reference'
    for name in (sys.argv[1:] or list(CASES)):
        run_case(name, **CASES[name])
