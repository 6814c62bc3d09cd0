"""Why does the bf16 stem BatchNorm bias gradient sit ~10 % off every reference?  Captures the stem's
tensors on the HIP path and recomputes its BatchNorm backward sums with torch in float64."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
import torch
import moco_util as U
from oracle.moco import MoCoOracle
from passl_amd.hip import nn as hnn, ops

K, N, HW = 1024, 8, 64
oracle = MoCoOracle(K=K, seed=0, t_max=200 * 5004, bf16=True)
model, opt, sched = U.build_product(K, torch.bfloat16)
U.load_oracle_state(model, oracle)
model.train()
gen = torch.Generator().manual_seed(1234)
xq = torch.randn(N, 3, HW, HW, generator=gen); xk = torch.randn(N, 3, HW, HW, generator=gen)
cap = {}
orig_bwd = ops.bn_bwd
def spy(dz, z, x, gamma, mean, invstd, dgamma, dbeta, **kw):
    if x.shape[-1] == 64 and 'stem' not in cap and x.shape[1] == HW // 2:
        cap['stem'] = dict(dz=dz.clone(), y=x.clone(), mean=mean.clone(), invstd=invstd.clone(),
                           scale=kw['scale'].clone(), shift=kw['shift'].clone(), relu=kw.get('relu'))
        before = dbeta.clone()
        r = orig_bwd(dz, z, x, gamma, mean, invstd, dgamma, dbeta, **kw)
        cap['stem']['dbeta'] = (dbeta - before).clone()
        return r
    return orig_bwd(dz, z, x, gamma, mean, invstd, dgamma, dbeta, **kw)
ops.bn_bwd = spy
out = U.product_step(model, opt, sched, xq.cuda(), xk.cuda())
ops.bn_bwd = orig_bwd
c = cap['stem']
dz, y = c['dz'].double(), c['y'].double()
mask = (y * c['scale'].double() + c['shift'].double()) > 0
g = torch.where(mask, dz, torch.zeros_like(dz))
ref_dbeta = g.sum(dim=(0, 1, 2))
print('relu mode', c['relu'], 'HIP dbeta vs recomputed from its own dz/y: rel', float((c['dbeta'].double() - ref_dbeta).norm() / ref_dbeta.norm()))
print('|dbeta| hip %.6e recomputed %.6e ; sum|g| %.4e ; |sum g| per channel mean %.4e' % (
    float(c['dbeta'].norm()), float(ref_dbeta.norm()), float(g.abs().sum()), float(ref_dbeta.abs().mean())))
# emulation on CPU with taps
ref = oracle.forward_backward(xq, xk)
gb = ref['grads']['0.bn1.bias']
print('emulated |dbeta| %.6e ; HIP-vs-emu rel %.4e' % (float(gb.norm()), float((c['dbeta'].cpu() - gb).norm() / gb.norm())))
print('per channel hip/emu ratio:', (c['dbeta'].cpu() / gb)[:16])
