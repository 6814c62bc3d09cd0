import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch, torch.nn.functional as F
import simsiam_util as U
from oracle import simsiam as S
from oracle import resnet50 as R
from passl_amd.hip import nn as hnn, config
from passl_amd.models import simsiam as PS
N, size = 8, 64
oracle = S.SimSiamOracle(seed=0, zero_init_residual=False, dtype=torch.float64, **U.SOLVER)
model, opt = U.build_product(torch.float32)
U.load_oracle_state(model, oracle)
model.train()
gen = torch.Generator().manual_seed(5)
x1 = torch.randn(N, 3, size, size, generator=gen); x2 = torch.randn(N, 3, size, size, generator=gen)
if os.environ.get("SWAP"): x1, x2 = x2, x1
# ---- oracle with retained intermediates
st = oracle.st
for n in S.trainable_keys(st): st[n] = st[n].detach().requires_grad_(True)
T = {}
def enc(x, tag):
    trunk = {'0.' + k[len('encoder.'):]: v for k, v in st.items() if k.startswith('encoder.') and not k.startswith('encoder.fc.')}
    ns = {}
    f = R.trunk_forward(trunk, x, False, ns, maxpool=True)
    f = F.adaptive_avg_pool2d(f, 1).flatten(1)
    def keep(n, t): t.retain_grad(); T[tag + n] = t; return t
    f = keep('f', f)
    a0 = keep('a0', f @ st['encoder.fc.0.weight']); h1 = keep('h1', F.relu(S.bn1d(st, 'encoder.fc.1', a0, ns)))
    a3 = keep('a3', h1 @ st['encoder.fc.3.weight']); h2 = keep('h2', F.relu(S.bn1d(st, 'encoder.fc.4', a3, ns)))
    a6 = keep('a6', h2 @ st['encoder.fc.6.weight'] + st['encoder.fc.6.bias']); z = keep('z', S.bn1d(st, 'encoder.fc.7', a6, ns))
    b0 = keep('b0', z @ st['predictor.0.weight']); g1 = keep('g1', F.relu(S.bn1d(st, 'predictor.1', b0, ns)))
    p = keep('p', g1 @ st['predictor.3.weight'] + st['predictor.3.bias'])
    return z, p
z1, p1 = enc(x1.double(), 'v1.'); z2, p2 = enc(x2.double(), 'v2.')
loss = -(S.cosine(p1, z2.detach()).mean() + S.cosine(p2, z1.detach()).mean()) * 0.5
loss.backward()
# ---- product with retained intermediates
P = {}
def mlp_fwd(self, x, tag):
    mods = list(self)
    def keep(n, t): t.retain_grad(); P[tag + n] = t; return t
    if len(mods) == 8:
        x = keep('f', x)
        a0 = keep('a0', mods[0](x, out_f32=True)); h1 = keep('h1', mods[1](a0, relu=True))
        a3 = keep('a3', mods[3](h1, out_f32=True)); h2 = keep('h2', mods[4](a3, relu=True))
        a6 = keep('a6', mods[6](h2, out_f32=True)); return keep('z', mods[7](a6))
    b0 = keep('b0', mods[0](x, out_f32=True)); g1 = keep('g1', mods[1](b0, relu=True))
    return keep('p', mods[3](g1, out_f32=True))
tagbox = ['v1.']
PS._MLP.forward = lambda self, x: mlp_fwd(self, x, tagbox[0])
orig_view = model._view
def view(x):
    r = orig_view(x); tagbox[0] = 'v2.'; return r
model._view = view
model.arena_q.clear_grad(); model.arena_p.clear_grad()
l = model([x1.cuda(), x2.cuda()])
l.backward(); torch.cuda.synchronize()
def rel(a, r): return float((a.double().cpu() - r).abs().max() / r.abs().max().clamp_min(1e-30))
print('loss', float(l), float(loss))
for tag in ('v1.', 'v2.'):
    for n in ('p', 'g1', 'b0', 'z', 'a6', 'h2', 'a3', 'h1', 'a0', 'f'):
        print(tag + n, 'fwd %.2e' % rel(P[tag + n].detach().reshape(T[tag + n].shape), T[tag + n].detach()), 'grad %.2e' % rel(P[tag + n].grad.reshape(T[tag + n].shape), T[tag + n].grad), ' |ref grad| %.2e' % float(T[tag+n].grad.abs().max()))
