"""A few R50 conv launches (fwd) for PMC stall breakdowns:  rocprofv3 --pmc ... -- python scratch/pmc_convs.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from passl_amd.hip import ops, plan as P
from passl_amd.hip.packer import WeightPacker
DEV, N, dtype = 'cuda', 256, torch.bfloat16
SHAPES = [(256, 256, 3, 1, 1, 14), (128, 128, 3, 1, 1, 28), (512, 512, 3, 1, 1, 7), (64, 64, 3, 1, 1, 56),
          (256, 1024, 1, 1, 0, 14), (1024, 256, 1, 1, 0, 14), (512, 2048, 1, 1, 0, 7), (2048, 512, 1, 1, 0, 7),
          (64, 256, 1, 1, 0, 56), (256, 64, 1, 1, 0, 56), (128, 512, 1, 1, 0, 28), (512, 128, 1, 1, 0, 28)]
for cin, cout, k, stv, pad, H in SHAPES:
    g = P.ConvGeom(cin, cout, k, stv, pad)
    fd = P.fwd_desc(g, N, H, H)
    wd = P.wgrad_desc(g, N, H, H)
    packer = WeightPacker()
    packer.add(0, cout, k, k, cin, fd.pack)
    w = torch.randn(cout * k * k * cin, device=DEV) * 0.05
    packer.build(DEV, dtype).run(w)
    x = torch.randn(N, H, H, cin, device=DEV).to(dtype)
    y = torch.empty(N, fd.OP, fd.OQ, cout, device=DEV, dtype=dtype)
    dy = torch.randn(N, fd.OP, fd.OQ, cout, device=DEV).to(dtype)
    dw = torch.zeros(cout, k * k * cin, device=DEV)
    for _ in range(3):
        ops.conv_igemm(fd, x, packer.view(fd.pack, cout), y)
        ops.conv_wgrad(wd, x, dy.view(-1, cout), dw)
    torch.cuda.synchronize()
print('done')
