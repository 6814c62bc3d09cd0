"""wgrad on dense (1x1 / Linear) shapes: wgrad_dma_kernel vs the 4-stage wgrad_pipe_kernel + exactness."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from passl_amd.hip import ops, plan as P, lib as L
from bench_convs_shapes import SHAPES
lib = L.load()
DEV = 'cuda'; N = 256; dtype = torch.bfloat16
def run(fn, iters=10):
    for _ in range(2): fn()
    s = torch.cuda.Event(True); e = torch.cuda.Event(True); s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / iters * 1e3
rows = [(cin, cout, k, st, pad, H, cnt, N) for cin, cout, k, st, pad, H, cnt in SHAPES if k == 1 and st == 1]
rows += [(768, 2304, 1, 1, 0, 1, 12, 12800), (768, 768, 1, 1, 0, 1, 12, 12800), (768, 3072, 1, 1, 0, 1, 12, 12800), (3072, 768, 1, 1, 0, 1, 12, 12800),
         (512, 1536, 1, 1, 0, 1, 8, 50432), (512, 2048, 1, 1, 0, 1, 8, 50432), (2048, 512, 1, 1, 0, 1, 8, 50432),
         (768, 2304, 1, 1, 0, 1, 12, 6400), (512, 2048, 1, 1, 0, 1, 12, 9856), (200, 136, 1, 1, 0, 1, 1, 1000), (72, 72, 1, 1, 0, 1, 1, 77)]
tot = {0: 0.0, 1: 0.0, 2: 0.0}
for cin, cout, k, st, pad, H, cnt, n in rows:
    g = P.ConvGeom(cin, cout, k, st, pad); wd = P.wgrad_desc(g, n, H, H)
    x = torch.randn(n, H, H, cin, device=DEV).to(dtype)
    dy = torch.randn(n * wd.OP * wd.OQ, cout, device=DEV).to(dtype)
    fl = 2.0 * n * wd.OP * wd.OQ * cout * cin
    res = {}
    outs = {}
    for mode in (0, 1, 2):
        lib.passl_hip_set_option(b'wgrad_pipe', mode)
        dw = torch.zeros(cout, cin, device=DEV)
        ops.conv_wgrad(wd, x, dy, dw)
        outs[mode] = dw.clone()
        res[mode] = run(lambda: ops.conv_wgrad(wd, x, dy, dw))
        if n == N: tot[mode] += res[mode] * cnt
    ref = dy.float().t() @ x.reshape(-1, cin).float()
    e0 = float((outs[0] - ref).abs().max() / ref.abs().max()); e1 = float((outs[1] - ref).abs().max() / ref.abs().max())
    e2 = float((outs[2] - ref).abs().max() / ref.abs().max())
    print('%4d->%4d @%3d n=%5d | dma %7.1f us (%4.0f TF) pipe4x32 %7.1f us (%4.0f TF) %+5.1f%% pipe2x64 %7.1f us (%4.0f TF) %+5.1f%% | err %.1e %.1e %.1e' % (
        cin, cout, H, n, res[0], fl / res[0] / 1e6, res[1], fl / res[1] / 1e6, (res[0] / res[1] - 1) * 100,
        res[2], fl / res[2] / 1e6, (res[0] / res[2] - 1) * 100, e0, e1, e2))
print('R50 dense 1x1 layers per pass: dma %.2f ms, pipe4x32 %.2f ms, pipe2x64 %.2f ms' % (tot[0] / 1e3, tot[1] / 1e3, tot[2] / 1e3))
