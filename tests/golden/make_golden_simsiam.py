"""Generate tests/golden/simsiam_*.npz by EXECUTING THE REFERENCE's v2 SimSiam sources (passl/models/simsiam.py
SimSiamPretain over passl/models/resnet.py; the paddle.vision ResNet they subclass is the tree's own copy,
passl_v110/modeling/backbones/resnetimagenet.py — see oracle/ref_runner_v2.load_simsiam) on torch-CPU through the
paddle shim; backward = torch autograd over the reference's forward graph, momentum SGD with the two parameter
groups of the yaml = oracle.simsiam.SimSiamOracle.apply_momentum (rule: passl/optimizer/momentum.py:150-158).

    python tests/golden/make_golden_simsiam.py          # its own process

Seed-defined inputs: weights oracle.simsiam.SimSiamOracle(seed=0, zero_init_residual=...); per step x1, x2 ~ N(0,1)
[N,3,S,S] from torch.Generator().manual_seed(777).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import ref_runner_v2                                         # noqa: E402
from oracle.simsiam import SimSiamOracle, trainable_keys, is_stat        # noqa: E402

CASES = {
    # residual branches switched on (gamma of every bn3 = 1): every conv gets a gradient
    'simsiam_r50_small': dict(N=32, S=64, steps=3, zero_init_residual=False),
    # the reference's own initialisation (zero_init_residual=True: every block starts as an identity)
    'simsiam_r50_zero_init': dict(N=32, S=64, steps=2, zero_init_residual=True),
}
SOLVER = dict(lr=2e-4, predictor_lr=5e-4, momentum=0.9, weight_decay=1e-4)
WATCH = ['encoder.conv1.weight', 'encoder.bn1.bias', 'encoder.layer1.0.conv2.weight', 'encoder.layer1.0.bn3.weight',
         'encoder.layer2.0.downsample.0.weight', 'encoder.layer3.5.conv1.weight', 'encoder.layer4.2.conv3.weight',
         'encoder.layer4.2.bn2.weight', 'encoder.fc.0.weight', 'encoder.fc.4.bias', 'encoder.fc.6.weight',
         'predictor.0.weight', 'predictor.1.weight', 'predictor.3.weight', 'predictor.3.bias']
STATS = ['encoder.bn1._mean', 'encoder.layer2.1.bn2._variance', 'encoder.layer4.2.bn3._mean', 'encoder.fc.1._variance',
         'encoder.fc.7._mean', 'predictor.1._variance']


def load_state(model, oracle):
    with torch.no_grad():
        sd = model.state_dict()
        assert set(sd.keys()) == set(oracle.st.keys()), set(sd.keys()) ^ set(oracle.st.keys())
        for k, t in oracle.st.items():
            assert sd[k].shape == t.shape, (k, sd[k].shape, t.shape)
            sd[k].copy_(t.detach())


def record(out, pre, loss, grads, oracle, z1, p1):
    out[pre + 'loss'] = np.float64(float(loss))
    for n in WATCH:
        out[pre + 'gradnorm/' + n] = np.float64(grads[n].double().norm().item())
        out[pre + 'pnorm/' + n] = np.float64(oracle.st[n].double().norm().item())
    for n in STATS:
        out[pre + 'stat/' + n] = oracle.st[n].detach().double().numpy()[:8].copy()
    out[pre + 'z1_head'] = z1[:, :8].double().numpy().copy()
    out[pre + 'p1_head'] = p1[:, :8].double().numpy().copy()


def run_case(ns, name, N, S, steps, zero_init_residual):
    torch.manual_seed(0)
    oracle = SimSiamOracle(seed=0, zero_init_residual=zero_init_residual, **SOLVER)
    model = ns.simsiam.simsiam_resnet50_pretrain()
    load_state(model, oracle)
    model.train()
    cap = {}
    enc_fwd = model.encoder.forward
    pred_fwd = model.predictor.forward
    model.encoder.forward = lambda x: cap.setdefault('z', []).append(enc_fwd(x)) or cap['z'][-1]
    model.predictor.forward = lambda x: cap.setdefault('p', []).append(pred_fwd(x)) or cap['p'][-1]
    gen = torch.Generator().manual_seed(777)
    out = {}
    for s in range(steps):
        x1 = torch.randn(N, 3, S, S, generator=gen)
        x2 = torch.randn(N, 3, S, S, generator=gen)
        for p in model.parameters():
            p.grad = None
        cap.clear()
        loss = model([x1, x2])
        loss.backward()
        ps = dict(model.named_parameters())
        grads = {n: ps[n].grad.detach().clone() for n in ps if ps[n].grad is not None}
        sd = model.state_dict()
        for k in oracle.st:
            oracle.st[k] = sd[k].detach().clone()
        assert set(grads) == set(trainable_keys(oracle.st)), set(grads) ^ set(trainable_keys(oracle.st))
        oracle.apply_momentum(grads)
        with torch.no_grad():
            for k in oracle.st:
                sd[k].copy_(oracle.st[k])
        record(out, 's%d_' % s, loss.item(), grads, oracle, cap['z'][0].detach(), cap['p'][0].detach())
        print(name, 'step', s, 'loss %.6f' % out['s%d_loss' % s])
    o64 = SimSiamOracle(seed=0, zero_init_residual=zero_init_residual, dtype=torch.float64, **SOLVER)
    gen = torch.Generator().manual_seed(777)
    for s in range(steps):
        x1 = torch.randn(N, 3, S, S, generator=gen)
        x2 = torch.randn(N, 3, S, S, generator=gen)
        r = o64.train_step(x1.double(), x2.double())
        record(out, 's%d_f64_' % s, r['loss'], r['grads'], o64, r['z1'], r['p1'])
    out['meta'] = np.array([N, S, steps, int(zero_init_residual)])
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)


if __name__ == '__main__':
    ns = ref_runner_v2.load_simsiam()
    only = sys.argv[1:]
    for name, c in CASES.items():
        if not only or name in only:
            run_case(ns, name, **c)
