"""Generate tests/golden/simclr_r18_*.npz by EXECUTING THE REFERENCE's BasicBlock trunk
(passl_v110/modeling/backbones/resnetcifar.py:41-118 block, :216-333 trunk without the stem max-pool) behind its
registered constructor ``ResNetsimclr(depth=18, frozen_stages=...)`` (resnetsimclr.py:25-91), with the neck / head of
configs/simclr/simclr_r18_cifar10.yaml:9-17 (NonLinearNeckfc3 512-512-128, SimCLRContrastiveHead T = 0.5), on torch-CPU
through the paddle shim (oracle/ref_runner.py).  The yaml's own backbone name, ``ResNetCifar``, is not registered in the
reference (tests/test_oracle_simclr.py pins the KeyError) — ``ResNetsimclr`` is the registered class over the same file.

Two cases: the trunk TRAINABLE (frozen_stages = -1: every BasicBlock kernel path runs forward and backward) and the
yaml's ``frozen_stages: 4`` (trunk on its running statistics, only the projector learns).  32 x 32 inputs (CIFAR).
Backward = torch autograd over the reference's forward graph; LARS = oracle.simclr.SimCLROracle.apply_lars.

    python tests/golden/make_golden_simclr_r18.py

Seed-defined inputs: weights oracle.simclr.SimCLROracle(seed=0, depth=18, in/hid 512, **SOLVER);
views torch.Generator().manual_seed(1818), per step x_q then x_k ~ N(0,1)."""
import copy
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import ref_runner                      # noqa: E402
from oracle.simclr import SimCLROracle             # noqa: E402

SOLVER = dict(T=0.5, lr=2.0, warmup_steps=2, t_max=1000)
# exclude_from_weight_decay as the yaml spells it (simclr_r18_cifar10.yaml:119): 'b_0' matches every bias's Paddle name
ORACLE_KW = dict(depth=18, in_channels=512, hid_channels=512, exclude=('scale', 'offset', 'b_0'))
MODEL_CFG = dict(
    name='SimCLR',
    backbone=dict(name='ResNetsimclr', depth=18, frozen_stages=-1),
    neck=dict(name='NonLinearNeckfc3', in_channels=512, hid_channels=512, out_channels=128, with_avg_pool=False),
    head=dict(name='SimCLRContrastiveHead', temperature=0.5),
)
CASES = {
    'simclr_r18_train': dict(N=16, hw=32, steps=3, frozen_stages=-1),
    'simclr_r18_frozen4': dict(N=16, hw=32, steps=3, frozen_stages=4),
}
WATCH = ['0.conv1.weight', '0.layer1.0.conv1.weight', '0.layer1.1.conv2.weight', '0.layer2.0.conv1.weight',
         '0.layer2.0.downsample.0.weight', '0.layer3.1.bn2.weight', '0.layer4.1.conv2.weight', '0.bn1.bias',
         '1.mlp.0.weight', '1.mlp.3.bias', '1.mlp.6.weight', '1.mlp.7.weight']
WATCH_STATS = ['0.bn1._mean', '0.layer2.0.downsample.1._variance', '0.layer4.1.bn2._mean', '1.mlp.7._variance']


def run_case(name, N, hw, steps, frozen_stages):
    torch.manual_seed(0)
    oracle = SimCLROracle(seed=0, **SOLVER, **ORACLE_KW)
    cfg = copy.deepcopy(MODEL_CFG)
    cfg['backbone']['frozen_stages'] = frozen_stages
    model = ref_runner.load().build_model(cfg)
    ref_runner.load_simclr_state(model, oracle)
    model.train()
    captured = {}
    model.head.register_forward_pre_hook(
        lambda mod, args: captured.update(q=args[0].detach().clone(), k=args[1].detach().clone()))
    gen = torch.Generator().manual_seed(1818)
    out = {}
    for s in range(steps):
        xq = torch.randn(N, 3, hw, hw, generator=gen)
        xk = torch.randn(N, 3, hw, hw, generator=gen)
        for p in model.parameters():
            p.grad = None
        res = model(xq, xk, mode='train')
        res['loss'].backward()
        psd = dict(model.encoder.named_parameters())
        grads = {n: psd[n].grad.detach().clone() for n in psd if psd[n].grad is not None}
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
            out[pre + 'gradnorm/' + n] = np.float64(grads[n].double().norm().item() if n in grads else 0.0)
            out[pre + 'pnorm/' + n] = np.float64(oracle.st[n].double().norm().item())
        for n in WATCH_STATS:
            out[pre + 'stat/' + n] = oracle.st[n][:8].numpy().astype(np.float64)
        print(name, 'step', s, 'loss %.6f acc1 %.3f lr %.4f trainable grads %d' % (
            out[pre + 'loss'], out[pre + 'acc1'], lr, len(grads)))
    out['meta'] = np.array([N, hw, steps, frozen_stages], dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)


if __name__ == '__main__':
    assert ref_runner.available(), 'needs /root/reference'
    for name in (sys.argv[1:] or list(CASES)):
        run_case(name, **CASES[name])
