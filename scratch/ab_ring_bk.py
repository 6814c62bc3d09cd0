"""ring igemm: BK=64 x 2 stages vs BK=32 x 4 stages on R50 (K >= 512) and ViT Linear shapes; bit-exactness."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from passl_amd.hip import ops, plan as P, lib as L
from passl_amd.hip.packer import WeightPacker
from bench_convs_shapes import SHAPES
lib = L.load()
DEV = 'cuda'; N = 256; dtype = torch.bfloat16
def run(fn, iters=10):
    for _ in range(2): fn()
    s = torch.cuda.Event(True); e = torch.cuda.Event(True); s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / iters * 1e3
rows = [(cin, cout, k, st, pad, H, cnt, N) for cin, cout, k, st, pad, H, cnt in SHAPES if k * k * cin >= 512]
rows += [(768, 2304, 1, 1, 0, 1, 12, 12800), (768, 768, 1, 1, 0, 1, 12, 12800), (768, 3072, 1, 1, 0, 1, 12, 12800), (3072, 768, 1, 1, 0, 1, 12, 12800),
         (512, 1536, 1, 1, 0, 1, 8, 50432), (512, 2048, 1, 1, 0, 1, 8, 50432), (2048, 512, 1, 1, 0, 1, 8, 50432),
         (768, 2304, 1, 1, 0, 1, 12, 6400), (512, 2048, 1, 1, 0, 1, 12, 9856), (576, 72, 1, 1, 0, 1, 1, 1000), (512, 200, 3, 2, 1, 9, 1, 5)]
tot = {64: [0.0, 0.0], 32: [0.0, 0.0]}
STAGES32 = int(os.environ.get('STAGES32', 4))
lib.passl_hip_set_option(b'igemm_ring_stages32', STAGES32)
lib.passl_hip_set_option(b'igemm_ring', 1); lib.passl_hip_set_option(b'igemm_ring_min_nk', 1)
for cin, cout, k, st, pad, H, cnt, n in rows:
    g = P.ConvGeom(cin, cout, k, st, pad); fd = P.fwd_desc(g, n, H, H); dds, _ = P.dgrad_plan(g, n, H, H)
    packer = WeightPacker()
    for d in [fd] + dds: packer.add(0, cout, k, k, cin, d.pack)
    packer.build(DEV, dtype).run(torch.randn(cout * k * k * cin, device=DEV) * 0.05)
    x = torch.randn(n, H, H, cin, device=DEV).to(dtype); y = torch.empty(n, fd.OP, fd.OQ, cout, device=DEV, dtype=dtype)
    dy = torch.randn(n, fd.OP, fd.OQ, cout, device=DEV).to(dtype); dx = torch.zeros(n, H, H, cin, device=DEV, dtype=dtype)
    fl = 2.0 * n * fd.OP * fd.OQ * cout * k * k * cin
    f = lambda: ops.conv_igemm(fd, x, packer.view(fd.pack, cout), y, relu=True)
    def dg():
        for d in dds: ops.conv_igemm(d, dy, packer.view(d.pack, cin), dx)
    res = {}; outs = {}
    for bk in (64, 32):
        lib.passl_hip_set_option(b'igemm_ring_bk', bk)
        f(); dg(); outs[bk] = (y.clone(), dx.clone())
        res[bk] = (run(f), run(dg) if cout % 64 == 0 else 0.0)
        if n == N: tot[bk][0] += res[bk][0] * cnt; tot[bk][1] += res[bk][1] * cnt
    same = torch.equal(outs[64][0], outs[32][0]) and torch.equal(outs[64][1], outs[32][1])
    print('%4d->%4d k%d s%d @%3d n=%5d | fwd bk64 %7.1f (%4.0f TF) bk32 %7.1f (%4.0f TF) %+5.1f%% | dgrad bk64 %7.1f bk32 %7.1f %+5.1f%% | identical %s' % (
        cin, cout, k, st, H, n, res[64][0], fl / res[64][0] / 1e6, res[32][0], fl / res[32][0] / 1e6, (res[64][0] / res[32][0] - 1) * 100,
        res[64][1], res[32][1], (res[64][1] / max(res[32][1], 1e-9) - 1) * 100, same))
print('R50 K>=512 layers per pass: fwd bk64 %.2f ms bk32 %.2f ms | dgrad bk64 %.2f ms bk32 %.2f ms' % (tot[64][0] / 1e3, tot[32][0] / 1e3, tot[64][1] / 1e3, tot[32][1] / 1e3))
