"""Interleaved A/B of wgrad variants (median of rounds): split counts, atomics off."""
import sys, os, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
os.environ['PASSL_WGRAD_DBG_DYNAMIC'] = '1'
import torch
from passl_amd.hip import ops, plan as P
libc = ctypes.CDLL(None)
def setdbg(v): libc.setenv(b'PASSL_WGRAD_DBG', str(v).encode(), 1)
DEV='cuda'; N=256; dtype=torch.bfloat16
SHAPES=[(64,64,3,1,1,56),(64,256,1,1,0,56),(128,512,1,1,0,28),(256,1024,1,1,0,14),(256,256,3,1,1,14),(512,2048,1,1,0,7),(512,512,3,1,1,7)]
def run(fn, iters=20):
    s=torch.cuda.Event(True); e=torch.cuda.Event(True); s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/iters*1e3
for cin,cout,k,st,pad,H in SHAPES:
    g=P.ConvGeom(cin,cout,k,st,pad); wd=P.wgrad_desc(g,N,H,H)
    x=torch.randn(N,H,H,cin,device=DEV).to(dtype)
    OP=(H+2*pad-k)//st+1
    dy=torch.randn(N*OP*OP,cout,device=DEV).to(dtype)
    dw=torch.zeros(cout,k*k*cin,device=DEV)
    M=N*OP*OP
    base=P.wgrad_splits(M,cout,k*k*cin,64)
    variants={'def(s=%d)'%base:(0,base),'noatomic':(1,base),'s/2':(0,max(1,base//2)),'s/4':(0,max(1,base//4)),'s/4 noat':(1,max(1,base//4)),'s/8':(0,max(1,base//8))}
    res={k:[] for k in variants}
    for r in range(5):
        for name,(dbg,sp) in variants.items():
            setdbg(dbg)
            res[name].append(run(lambda: ops.conv_wgrad(wd,x,dy,dw,sp)))
    setdbg(0)
    print('%4d->%4d k%d @%2d: ' % (cin,cout,k,H) + '  '.join('%s %.1f' % (k, sorted(v)[len(v)//2]) for k,v in res.items()))
