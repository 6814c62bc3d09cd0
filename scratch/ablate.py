import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from passl_amd.hip import ops, plan as P
from passl_amd.hip.packer import WeightPacker
DEV='cuda'; N=256; dtype=torch.bfloat16
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s=torch.cuda.Event(True); e=torch.cuda.Event(True); s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/iters*1e3
for cin,cout,k,st,pad,H in [(64,256,1,1,0,56),(256,64,1,1,0,56),(64,64,1,1,0,56),(128,512,1,1,0,28),(64,64,3,1,1,56)]:
    g=P.ConvGeom(cin,cout,k,st,pad); fd=P.fwd_desc(g,N,H,H)
    packer=WeightPacker(); packer.add(0,cout,k,k,cin,fd.pack)
    packer.build(DEV,dtype).run(torch.randn(cout*k*k*cin,device=DEV)*0.05)
    x=torch.randn(N,H,H,cin,device=DEV).to(dtype); y=torch.empty(N,fd.OP,fd.OQ,cout,device=DEV,dtype=dtype)
    print('%d->%d k%d @%d: %.1f us' % (cin,cout,k,H,timeit(lambda: ops.conv_igemm(fd,x,packer.view(fd.pack,cout),y))))
