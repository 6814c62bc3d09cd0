"""BatchNorm streaming kernels at the ResNet-50 bs256 shapes, per bn_stream_unroll variant
(0 = grid-stride kernels, 2/4/8 = tile form).  python scratch/bench_bn.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from passl_amd.hip import lib as L, ops

lib = L.load()
dev = 'cuda'
SHAPES = [(256 * 112 * 112, 64, 1), (256 * 56 * 56, 64, 6), (256 * 56 * 56, 256, 4), (256 * 28 * 28, 128, 8),
          (256 * 28 * 28, 512, 5), (256 * 14 * 14, 256, 12), (256 * 14 * 14, 1024, 7), (256 * 7 * 7, 512, 6),
          (256 * 7 * 7, 2048, 4)]


def timeit(f, n=10):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        f()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def main():
    st = L.stream()
    tot = {}
    print('%-22s %s' % ('shape x count', '  '.join('U=%d: apply us TB/s | bwd us TB/s' % u for u in (0, 2, 4, 8))))
    for M, C, cnt in SHAPES:
        # rotate over several buffers so that the 256 MB MALL does not serve the reads
        nbuf = max(2, int(1.2e9 // (M * C * 2)) + 1)
        nbuf = min(nbuf, 8)
        xs = [torch.randn(M, C, device=dev).bfloat16() for _ in range(nbuf)]
        gs = [torch.randn(M, C, device=dev).bfloat16() for _ in range(nbuf)]
        zs = [torch.empty(M, C, device=dev, dtype=torch.bfloat16) for _ in range(nbuf)]
        mask = torch.empty(M * C // 8, dtype=torch.uint8, device=dev)
        scale = torch.rand(C, device=dev) + 0.5
        shift = torch.randn(C, device=dev)
        coef = torch.randn(3 * C, device=dev)
        row = '%9d x %4d x%2d  ' % (M, C, cnt)
        for U in (0, 2, 4, 8):
            L.check(lib.passl_hip_set_option(b'bn_stream_unroll', U), 'opt')
            k = [0]

            def fa():
                i = k[0] % nbuf
                k[0] += 1
                L.check(lib.passl_hip_bn_apply(L.ptr(xs[i]), L.ptr(scale), L.ptr(shift), None, L.ptr(zs[i]),
                                               L.ptr(mask), M, C, 1, L.dt(xs[i]), st), 'a')

            def fb():
                i = k[0] % nbuf
                k[0] += 1
                L.check(lib.passl_hip_bn_bwd_apply(L.ptr(gs[i]), None, L.ptr(xs[i]), L.ptr(coef), None, None,
                                                   L.ptr(zs[i]), None, M, C, 0, L.dt(xs[i]), st), 'b')
            ta, tb = timeit(fa), timeit(fb)
            ba, bb = M * C * (4 + 0.125), M * C * 6
            row += '| %7.1f %5.2f  %7.1f %5.2f ' % (ta, ba / ta / 1e6, tb, bb / tb / 1e6)
            tot[U] = tot.get(U, 0) + cnt * (ta + tb)
        print(row)
        del xs, gs, zs
    print('weighted total per step (us):', {u: round(v) for u, v in tot.items()})


main()
