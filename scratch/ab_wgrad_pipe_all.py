"""wgrad on every R50 conv shape: wgrad_dma_kernel (wgrad_pipe=0) vs wgrad_pipe_kernel 2 x 64 (wgrad_pipe=2)."""
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
tot = {0: 0.0, 2: 0.0}; fl_tot = 0.0
for cin, cout, k, st, pad, H, cnt in SHAPES:
    g = P.ConvGeom(cin, cout, k, st, pad); wd = P.wgrad_desc(g, N, H, H)
    x = torch.randn(N, H, H, cin, device=DEV).to(dtype)
    dy = torch.randn(N * wd.OP * wd.OQ, cout, device=DEV).to(dtype)
    fl = 2.0 * N * wd.OP * wd.OQ * cout * k * k * cin
    res = {}; outs = {}
    for mode in (0, 2):
        lib.passl_hip_set_option(b'wgrad_pipe', mode)
        dw = torch.zeros(cout, k * k * cin, device=DEV)
        ops.conv_wgrad(wd, x, dy, dw)
        outs[mode] = dw.clone()
        res[mode] = run(lambda: ops.conv_wgrad(wd, x, dy, dw))
        tot[mode] += res[mode] * cnt
    fl_tot += fl * cnt
    err = float((outs[0] - outs[2]).abs().max() / outs[0].abs().max())
    print('%4d->%4d k%d s%d @%3d x%d | dma %7.1f us (%4.0f TF) pipe %7.1f us (%4.0f TF) %+5.1f%% | rel diff %.1e' % (
        cin, cout, k, st, H, cnt, res[0], fl / res[0] / 1e6, res[2], fl / res[2] / 1e6, (res[0] / res[2] - 1) * 100, err))
print('R50 wgrad per pass: dma %.2f ms (%.0f TF), pipe %.2f ms (%.0f TF)' % (tot[0] / 1e3, fl_tot / tot[0] / 1e6, tot[2] / 1e3, fl_tot / tot[2] / 1e6))
