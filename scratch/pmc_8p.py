"""A few launches of ONE implicit-GEMM shape with a forced kernel choice, for rocprofv3 PMC passes.
   python scratch/pmc_8p.py M K N mode(0 ring / 2 8p) direct(0/1) [iters]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from passl_amd.hip import ops, plan as P, lib as L
from passl_amd.hip.packer import WeightPacker
lib = L.load(); DEV = 'cuda'; dtype = torch.bfloat16
M, cin, cout, mode, direct = (int(a) for a in sys.argv[1:6])
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 5
lib.passl_hip_set_option(b'igemm_8p', mode); lib.passl_hip_set_option(b'igemm_8p_direct', direct)
g = P.ConvGeom(cin, cout, 1, 1, 0); fd = P.fwd_desc(g, M, 1, 1)
packer = WeightPacker(); packer.add(0, cout, 1, 1, cin, fd.pack)
packer.build(DEV, dtype).run(torch.randn(cout * cin, device=DEV) * 0.05)
x = torch.randn(M, cin, device=DEV).to(dtype); y = torch.empty(M, cout, device=DEV, dtype=dtype)
for _ in range(iters):
    ops.conv_igemm(fd, x, packer.view(fd.pack, cout), y)
torch.cuda.synchronize()
print('done kernel', lib.passl_hip_last_igemm_kernel())
