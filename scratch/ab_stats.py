"""Interleaved A/B of the igemm epilogue variants (median of rounds)."""
import sys, os, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
os.environ['PASSL_IGEMM_DBG_DYNAMIC'] = '1'
import torch
from passl_amd.hip import ops, plan as P
from passl_amd.hip.packer import WeightPacker
libc = ctypes.CDLL(None)
def setdbg(v): libc.setenv(b'PASSL_IGEMM_DBG', str(v).encode(), 1)
DEV='cuda'; N=256; dtype=torch.bfloat16
SHAPES=[(64,64,1,1,0,56),(64,64,3,1,1,56),(64,256,1,1,0,56),(256,64,1,1,0,56),(128,512,1,1,0,28),(256,1024,1,1,0,14),(256,256,3,1,1,14)]
def run(fn, iters=20):
    s=torch.cuda.Event(True); e=torch.cuda.Event(True); s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/iters*1e3
for cin,cout,k,st,pad,H in SHAPES:
    g=P.ConvGeom(cin,cout,k,st,pad); fd=P.fwd_desc(g,N,H,H)
    packer=WeightPacker(); packer.add(0,cout,k,k,cin,fd.pack)
    packer.build(DEV,dtype).run(torch.randn(cout*k*k*cin,device=DEV)*0.05)
    x=torch.randn(N,H,H,cin,device=DEV).to(dtype); y=torch.empty(N,fd.OP,fd.OQ,cout,device=DEV,dtype=dtype)
    stt=ops.conv_stats_buffer(N*fd.OP*fd.OQ,cout,DEV)
    st1=torch.zeros(1,cout,2,device=DEV)
    variants={'base':(0,None),'stats':(0,stt),'stats_noatomic':(16,stt),'stats_R1':(0,st1)}
    res={k:[] for k in variants}
    for r in range(7):
        for name,(dbg,sb) in variants.items():
            setdbg(dbg)
            res[name].append(run(lambda: ops.conv_igemm(fd,x,packer.view(fd.pack,cout),y,stats=sb)))
    setdbg(0)
    print('%4d->%4d k%d @%2d: ' % (cin,cout,k,H) + '  '.join('%s %.1f' % (k, sorted(v)[len(v)//2]) for k,v in res.items()))
