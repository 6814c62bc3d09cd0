"""wgrad on R50 + ViT shapes: full kernel vs PASSL_WGRAD_DBG=1 (epilogue atomics skipped)."""
import sys, os
os.environ['PASSL_WGRAD_DBG_DYNAMIC'] = '1'
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from passl_amd.hip import ops, plan as P
from bench_convs_shapes import SHAPES
DEV = 'cuda'; N = 256; dtype = torch.bfloat16
def run(fn, iters=10):
    for _ in range(2): fn()
    s = torch.cuda.Event(True); e = torch.cuda.Event(True); s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / iters * 1e3
tot = [0.0, 0.0, 0.0]
rows = [(cin, cout, k, st, pad, H, cnt, N) for cin, cout, k, st, pad, H, cnt in SHAPES]
rows += [(768, 2304, 1, 1, 0, 1, 12, 12800), (768, 3072, 1, 1, 0, 1, 12, 12800), (3072, 768, 1, 1, 0, 1, 12, 12800),
         (512, 2048, 1, 1, 0, 1, 8, 50432), (2048, 512, 1, 1, 0, 1, 8, 50432)]
for cin, cout, k, st, pad, H, cnt, n in rows:
    g = P.ConvGeom(cin, cout, k, st, pad); wd = P.wgrad_desc(g, n, H, H)
    x = torch.randn(n, H, H, cin, device=DEV).to(dtype)
    dy = torch.randn(n * wd.OP * wd.OQ, cout, device=DEV).to(dtype)
    dw = torch.zeros(cout, k * k * cin, device=DEV)
    fl = 2.0 * n * wd.OP * wd.OQ * cout * k * k * cin
    f = lambda: ops.conv_wgrad(wd, x, dy, dw)
    os.environ['PASSL_WGRAD_DBG'] = '0'; t0 = run(f)
    os.environ['PASSL_WGRAD_DBG'] = '1'; t1 = run(f)
    sp = P.wgrad_splits(n * wd.OP * wd.OQ, cout, k * k * cin, 64)
    if n == N:
        tot[0] += t0 * cnt; tot[1] += t1 * cnt; tot[2] += fl * cnt
    print('%4d->%4d k%d s%d @%3d n=%5d x%-2d splits=%2d | full %7.1f us (%4.0f TF)  no-atomics %7.1f us (%4.0f TF)' % (
        cin, cout, k, st, H, n, cnt, sp, t0, fl / t0 / 1e6, t1, fl / t1 / 1e6))
print('R50 pass: full %.2f ms (%.0f TF), no atomics %.2f ms (%.0f TF)' % (tot[0] / 1e3, tot[2] / tot[0] / 1e6, tot[1] / 1e3, tot[2] / tot[1] / 1e6))
