import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from passl_amd.hip import ops, plan as P, lib as L
from passl_amd.hip.packer import WeightPacker
lib = L.load(); DEV='cuda'; dtype=torch.bfloat16
lib.passl_hip_set_option(b'igemm_ring_min_nk', 1)
def cmp(name, fn, out):
    lib.passl_hip_set_option(b'igemm_ring', 0); out.zero_(); fn(); a = out.clone()
    res = []
    for bm in (128, 256):
        lib.passl_hip_set_option(b'igemm_ring', 1); lib.passl_hip_set_option(b'igemm_ring_bm', bm); out.zero_(); fn(); b = out.clone()
        d = (a.float() - b.float()).abs()
        res.append('bm%d: maxdiff %.3e nnz %d/%d' % (bm, d.max().item(), int((d > 0).sum()), d.numel()))
    print(name, ' | '.join(res))
for (cin,cout,k,st,pad,H,N) in [(64,64,3,1,1,56,4),(128,128,3,2,1,28,3),(256,256,3,1,1,14,8),(512,128,1,1,0,7,2),(256,512,1,2,0,14,3),(64,256,1,1,0,56,2),(1024,256,1,1,0,14,32),(512,512,3,1,1,7,32)]:
    g=P.ConvGeom(cin,cout,k,st,pad); fd=P.fwd_desc(g,N,H,H); dds,sk=P.dgrad_plan(g,N,H,H)
    packer=WeightPacker()
    for d in [fd]+dds: packer.add(0,cout,k,k,cin,d.pack)
    packer.build(DEV,dtype).run(torch.randn(cout*k*k*cin,device=DEV)*0.05)
    x=torch.randn(N,H,H,cin,device=DEV).to(dtype); y=torch.zeros(N,fd.OP,fd.OQ,cout,device=DEV,dtype=dtype)
    dy=torch.randn(N,fd.OP,fd.OQ,cout,device=DEV).to(dtype); dx=torch.zeros(N,H,H,cin,device=DEV,dtype=dtype)
    cmp('fwd  %s' % ((cin,cout,k,st,H,N),), lambda: ops.conv_igemm(fd,x,packer.view(fd.pack,cout),y), y)
    def dg():
        for d in dds: ops.conv_igemm(d,dy,packer.view(d.pack,cin),dx)
    cmp('dgrad%s' % ((cin,cout,k,st,H,N),), dg, dx)
    sc=torch.rand(cout,device=DEV)+0.5; sh=torch.randn(cout,device=DEV); res=torch.randn_like(y)
    cmp('fwd+epi', lambda: ops.conv_igemm(fd,x,packer.view(fd.pack,cout),y,scale=sc,shift=sh,residual=res,relu=True), y)
