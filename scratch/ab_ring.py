"""Interleaved A/B: igemm_kernel vs igemm_ring_kernel per R50 layer (fwd and dgrad), bs 256 bf16."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from passl_amd.hip import ops, plan as P, lib as L
from passl_amd.hip.packer import WeightPacker
from bench_convs_shapes import SHAPES
lib = L.load()
DEV='cuda'; N=int(os.environ.get('BATCH',256)); dtype=torch.bfloat16
def run(fn, iters=10):
    s=torch.cuda.Event(True); e=torch.cuda.Event(True); s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/iters*1e3
tot={'fwd_old':0,'fwd_r256':0,'fwd_r128':0,'dg_old':0,'dg_r256':0,'dg_r128':0}
for cin,cout,k,st,pad,H,cnt in SHAPES:
    g=P.ConvGeom(cin,cout,k,st,pad); fd=P.fwd_desc(g,N,H,H); dds,_=P.dgrad_plan(g,N,H,H)
    packer=WeightPacker()
    for d in [fd]+dds: packer.add(0,cout,k,k,cin,d.pack)
    packer.build(DEV,dtype).run(torch.randn(cout*k*k*cin,device=DEV)*0.05)
    x=torch.randn(N,H,H,cin,device=DEV).to(dtype); y=torch.empty(N,fd.OP,fd.OQ,cout,device=DEV,dtype=dtype)
    dy=torch.randn(N,fd.OP,fd.OQ,cout,device=DEV).to(dtype); dx=torch.zeros(N,H,H,cin,device=DEV,dtype=dtype)
    fl=2.0*N*fd.OP*fd.OQ*cout*k*k*cin
    f=lambda: ops.conv_igemm(fd,x,packer.view(fd.pack,cout),y)
    def dg():
        for d in dds: ops.conv_igemm(d,dy,packer.view(d.pack,cin),dx)
    res={k2:[] for k2 in tot}
    for r in range(5):
        lib.passl_hip_set_option(b'igemm_ring',0); res['fwd_old'].append(run(f)); res['dg_old'].append(run(dg))
        lib.passl_hip_set_option(b'igemm_ring',1); lib.passl_hip_set_option(b'igemm_ring_min_tiles',1)
        lib.passl_hip_set_option(b'igemm_ring_bm',256)
        res['fwd_r256'].append(run(f)); res['dg_r256'].append(run(dg))
        lib.passl_hip_set_option(b'igemm_ring_bm',128)
        res['fwd_r128'].append(run(f)); res['dg_r128'].append(run(dg))
    med={k2:sorted(v)[len(v)//2] for k2,v in res.items()}
    for k2 in tot: tot[k2]+=med[k2]*cnt
    tiles=((N*fd.OP*fd.OQ+255)//256)*((cout+127)//128 if cout>64 else 1)
    print('%4d->%4d k%d s%d @%3d x%d tiles256=%5d | fwd old %6.1f (%4.0f TF) r256 %6.1f (%4.0f TF) r128 %6.1f (%4.0f TF) | dgrad old %6.1f r256 %6.1f r128 %6.1f' % (cin,cout,k,st,H,cnt,tiles,med['fwd_old'],fl/med['fwd_old']/1e6,med['fwd_r256'],fl/med['fwd_r256']/1e6,med['fwd_r128'],fl/med['fwd_r128']/1e6,med['dg_old'],med['dg_r256'],med['dg_r128']))
print({k2: round(v/1e3,3) for k2,v in tot.items()}, 'ms per pass')
