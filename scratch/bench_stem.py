"""Stem conv (7x7 s2, 3->64) at bs 256, 224^2: spatially tiled kernel vs the generic implicit GEMM."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from passl_amd.hip import ops, plan as P, lib as L
from passl_amd.hip.packer import WeightPacker
DEV, dtype, N = 'cuda', torch.bfloat16, 256
lib = L.load()
d = P.stem_desc(64, N, 224, 224)
_Hp, Wp = P.stem_padded_hw(224, 224)
x = torch.randn(N, 3, 224, 224, device=DEV)
xp = ops.nchw_to_nhwc_pad(x, P.STEM_PAD, Wp, P.STEM_CP, dtype)
packer = WeightPacker(); packer.add(0, 64, 7, 7, 3, d.pack)
packer.build(DEV, dtype).run(torch.randn(64 * 7 * 7 * 3, device=DEV) * 0.05)
ys = [torch.empty(N, 112, 112, 64, dtype=dtype, device=DEV) for _ in range(3)]
slab, tiles = ops.conv_stats_buffer(d, DEV)
sc, sh = torch.rand(64, device=DEV) + 0.5, torch.randn(64, device=DEV)
def timeit(f, n=10):
    for _ in range(3): f(0)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(n): f(i)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for on in (0, 1):
    lib.passl_hip_set_option(b'stem_kernel', on)
    a = timeit(lambda i: ops.conv_igemm(d, xp, packer.view(d.pack, 64), ys[i % 3]))
    b = timeit(lambda i: ops.conv_igemm(d, xp, packer.view(d.pack, 64), ys[i % 3], stats=slab))
    c = timeit(lambda i: ops.conv_igemm(d, xp, packer.view(d.pack, 64), ys[i % 3], scale=sc, shift=sh, relu=True))
    print('stem_kernel=%d  plain %.1f us  +stats %.1f us  +affine/relu %.1f us   (520 MB: %.2f TB/s plain)' % (on, a, b, c, 520e6 / a / 1e6))
