"""Which part of the MoCo step can torch.cuda.graph capture on this ROCm build?  python scratch/graph_probe.py <case>
cases: fwd | fwdbwd | step   (env PASSL_OVERLAP / PASSL_FORK_DOWNSAMPLE select the side-stream variants)"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
import torch
import moco_util as U
case = sys.argv[1]
dtype = torch.bfloat16
model, opt, sched = U.build_product(512, dtype)
model.train()
N = 16
xq = torch.randn(N, 3, 64, 64, device='cuda'); xk = torch.randn(N, 3, 64, 64, device='cuda')

def fwd():
    with torch.no_grad():
        return model.encoder_k(xq)

def fwdbwd():
    out = model(xq, xk, mode='train')
    opt.clear_grad()
    out['loss'].backward()
    return out['loss']

def step():
    out = model(xq, xk, mode='train')
    opt.clear_grad()
    out['loss'].backward()
    opt.step()
    return out['loss']

fn = dict(fwd=fwd, fwdbwd=fwdbwd, step=step)[case]
for _ in range(2):
    r = fn()
torch.cuda.synchronize()
print(case, 'eager ok', float(r.float().sum()), flush=True)
if case == 'step':
    opt.push_hyper()
from passl_amd.hip import streams
streams.reset()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    r = fn()
print(case, 'captured', flush=True)
g.replay(); torch.cuda.synchronize()
print(case, 'replayed', float(r.float().sum()), flush=True)
