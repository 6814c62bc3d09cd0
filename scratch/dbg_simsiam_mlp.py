import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch, torch.nn.functional as F
import simsiam_util as U
from oracle import simsiam as S
from passl_amd.hip import nn as hnn
oracle = S.SimSiamOracle(seed=0, zero_init_residual=False, **U.SOLVER)
model, opt = U.build_product(torch.float32)
U.load_oracle_state(model, oracle)
model.train()
gen = torch.Generator().manual_seed(1)
f = torch.randn(8, 2048, generator=gen)
dz = torch.randn(8, 2048, generator=gen)
st = {k: v.detach().clone().double() for k, v in oracle.st.items()}
# oracle chain with retained grads
fr = f.double().requires_grad_(True)
ns = {}
t = {}
def keep(n, x): x.retain_grad(); t[n] = x; return x
a0 = keep('a0', fr @ st['encoder.fc.0.weight'])
h1 = keep('h1', F.relu(S.bn1d(st, 'encoder.fc.1', a0, ns)))
a3 = keep('a3', h1 @ st['encoder.fc.3.weight'])
h2 = keep('h2', F.relu(S.bn1d(st, 'encoder.fc.4', a3, ns)))
a6 = keep('a6', h2 @ st['encoder.fc.6.weight'] + st['encoder.fc.6.bias'])
z = S.bn1d(st, 'encoder.fc.7', a6, ns)
z.backward(dz.double())
# product chain
mods = list(model.encoder.fc)
fx = f.cuda().requires_grad_(True)
p = {}
def keepp(n, x): x.retain_grad(); p[n] = x; return x
model.arena_q.clear_grad()
b0 = keepp('a0', mods[0](fx, out_f32=True))
g1 = keepp('h1', mods[1](b0, relu=True))
b3 = keepp('a3', mods[3](g1, out_f32=True))
g2 = keepp('h2', mods[4](b3, relu=True))
b6 = keepp('a6', mods[6](g2, out_f32=True))
zz = mods[7](b6)
zz.backward(dz.cuda())
torch.cuda.synchronize()
def rel(a, r): return float((a.double().cpu() - r).abs().max() / r.abs().max())
print('fwd z', rel(zz.detach(), z.detach()))
for n in ('a6', 'h2', 'a3', 'h1', 'a0'):
    print(n, 'fwd', rel(p[n].detach(), t[n].detach()), 'grad', rel(p[n].grad, t[n].grad))
print('dx', rel(fx.grad, fr.grad))
d = (p['h2'].grad.double().cpu() - t['h2'].grad)
print('h2 grad err: per-column std over rows', float(d.std(0).mean()), 'per-column mean abs', float(d.mean(0).abs().mean()), 'max', float(d.abs().max()), 'ref max', float(t['h2'].grad.abs().max()))
