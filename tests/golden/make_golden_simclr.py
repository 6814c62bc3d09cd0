"""Generate tests/golden/simclr_*.npz by EXECUTING THE REFERENCE's SimCLR sources
(passl_v110/modeling/architectures/simclr.py, heads/simclr_contrastive_head.py,
backbones/resnetsimclr.py + resnetcifar.py, necks/base_neck.py:NonLinearNeckfc3) on torch-CPU
through the paddle shim (oracle/ref_runner.py).  Backward = torch autograd over the reference's
forward graph; the LARS update is oracle.simclr.SimCLROracle.apply_lars (Paddle's optimizer kernel
is not in the reference tree).  Run in the build container:

    python tests/golden/make_golden_simclr.py

Seed-defined inputs (regenerable on the GPU box without /root/reference):
    weights : oracle.simclr.SimCLROracle(seed=0, **SOLVER)
    views   : torch.Generator().manual_seed(4321); per step x_q then x_k ~ N(0,1)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import ref_runner                      # noqa: E402
from oracle.simclr import SimCLROracle             # noqa: E402

# lr schedule compressed so that the 2nd/3rd step see a non-zero LARS step (the yaml's warm-up
# starts at lr = 0: the first update of the reference changes nothing but the BN statistics)
SOLVER = dict(T=0.1, lr=4.0, warmup_steps=2, t_max=1000)
CASES = {
    'simclr_r50_small': dict(N=8, hw=64, steps=3),
    'simclr_r50_b32': dict(N=32, hw=96, steps=2),
}
WATCH = ['0.conv1.weight', '0.layer1.0.conv2.weight', '0.layer2.0.downsample.0.weight',
         '0.layer4.2.conv3.weight', '0.layer3.5.bn2.weight', '0.bn1.bias',
         '1.mlp.0.weight', '1.mlp.3.bias', '1.mlp.6.weight', '1.mlp.7.weight']
WATCH_STATS = ['0.bn1._mean', '0.bn1._variance', '1.mlp.7._mean', '1.mlp.7._variance']


def views(gen, N, hw):
    return torch.randn(N, 3, hw, hw, generator=gen), torch.randn(N, 3, hw, hw, generator=gen)


def run_case(name, N, hw, steps):
    torch.manual_seed(0)
    oracle = SimCLROracle(seed=0, **SOLVER)
    model = ref_runner.build_reference_simclr(T=SOLVER['T'])
    ref_runner.load_simclr_state(model, oracle)
    model.train()
    captured = {}
    model.head.register_forward_pre_hook(
        lambda mod, args: captured.update(q=args[0].detach().clone(), k=args[1].detach().clone()))
    gen = torch.Generator().manual_seed(4321)
    out = {}
    for s in range(steps):
        xq, xk = views(gen, N, hw)
        for p in model.parameters():
            p.grad = None
        res = model(xq, xk, mode='train')
        res['loss'].backward()
        psd = dict(model.encoder.named_parameters())
        grads = {n: psd[n].grad.detach().clone() for n in psd if psd[n].requires_grad}
        oracle.st = {n: p.detach().clone() for n, p in model.encoder.state_dict().items()}
        lr = oracle.lr()
        oracle.apply_lars(grads)
        with torch.no_grad():
            for n, p in model.encoder.state_dict().items():
                p.copy_(oracle.st[n])
        pre = 's%d_' % s
        q, k = captured['q'], captured['k']
        out[pre + 'loss'] = np.float64(res['loss'].item())
        out[pre + 'acc1'] = np.float64(float(res['acc1']))
        out[pre + 'lr'] = np.float64(lr)
        out[pre + 'q_head'] = q[:, :8].numpy().copy()
        out[pre + 'k_head'] = k[:, :8].numpy().copy()
        out[pre + 'ab_head'] = (q @ k.t() / SOLVER['T'])[:, :8].numpy().copy()
        for n in WATCH:
            out[pre + 'gradnorm/' + n] = np.float64(grads[n].double().norm().item())
            out[pre + 'pnorm/' + n] = np.float64(oracle.st[n].double().norm().item())
        for n in WATCH_STATS:
            out[pre + 'stat/' + n] = oracle.st[n][:8].numpy().astype(np.float64)
        print(name, 'step', s, 'loss %.6f acc1 %.3f lr %.4f' % (out[pre + 'loss'], out[pre + 'acc1'], lr))
    # same steps in float64 (oracle): the conditioning yardstick used by the GPU parity bounds
    o64 = SimCLROracle(seed=0, **SOLVER)
    for n in o64.st:
        o64.st[n] = o64.st[n].double()
    gen = torch.Generator().manual_seed(4321)
    for s in range(steps):
        xq, xk = views(gen, N, hw)
        r = o64.train_step(xq.double(), xk.double())
        pre = 's%d_f64_' % s
        out[pre + 'loss'] = np.float64(float(r['loss']))
        out[pre + 'q_head'] = r['q'][:, :8].numpy().copy()
        out[pre + 'ab_head'] = (r['q'] @ r['k'].t() / SOLVER['T'])[:, :8].numpy().copy()
        for n in WATCH:
            out[pre + 'gradnorm/' + n] = np.float64(r['grads'][n].norm().item())
            out[pre + 'pnorm/' + n] = np.float64(o64.st[n].norm().item())
        for n in WATCH_STATS:
            out[pre + 'stat/' + n] = o64.st[n][:8].numpy().copy()
        print(name, 'f64 step', s, 'loss %.6f' % out[pre + 'loss'])
    out['meta'] = np.array([N, hw, steps], dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)


if __name__ == '__main__':
    assert ref_runner.available(), 'needs /root/reference'
    for name in (sys.argv[1:] or list(CASES)):
        run_case(name, **CASES[name])
