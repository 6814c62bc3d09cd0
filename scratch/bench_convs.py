"""Per-shape microbenchmark of the conv kernels (fwd / dgrad / wgrad) at bs 256, bf16."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from passl_amd.hip import ops, plan as P
from passl_amd.hip.packer import WeightPacker

DEV = 'cuda'
# PASSL_OPTS="igemm_ring_bk=32,igemm_8p=0": library options for A/B runs
for kv in filter(None, os.environ.get('PASSL_OPTS', '').split(',')):
    k, v = kv.split('=')
    from passl_amd.hip import lib as _L
    rc = _L.load().passl_hip_set_option(k.encode(), int(v))
    assert rc == 0, (k, v, rc)
N = int(os.environ.get('BATCH', 256))
dtype = torch.bfloat16 if os.environ.get('DTYPE', 'bf16') == 'bf16' else torch.float32
# (cin, cout, k, stride, pad, H, count per forward pass)
SHAPES = [
    (64, 64, 1, 1, 0, 56, 1), (64, 64, 3, 1, 1, 56, 3), (64, 256, 1, 1, 0, 56, 4), (256, 64, 1, 1, 0, 56, 2),
    (256, 128, 1, 1, 0, 56, 1), (128, 128, 3, 2, 1, 56, 1), (128, 512, 1, 1, 0, 28, 4), (256, 512, 1, 2, 0, 56, 1),
    (512, 128, 1, 1, 0, 28, 3), (128, 128, 3, 1, 1, 28, 3),
    (512, 256, 1, 1, 0, 28, 1), (256, 256, 3, 2, 1, 28, 1), (256, 1024, 1, 1, 0, 14, 6), (512, 1024, 1, 2, 0, 28, 1),
    (1024, 256, 1, 1, 0, 14, 5), (256, 256, 3, 1, 1, 14, 5),
    (1024, 512, 1, 1, 0, 14, 1), (512, 512, 3, 2, 1, 14, 1), (512, 2048, 1, 1, 0, 7, 3), (1024, 2048, 1, 2, 0, 14, 1),
    (2048, 512, 1, 1, 0, 7, 2), (512, 512, 3, 1, 1, 7, 2),
]

def timeit(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters

tot = {'fwd': [0, 0], 'dgrad': [0, 0], 'wgrad': [0, 0]}
print('%-28s %8s | %8s %7s | %8s %7s | %8s %7s' % ('shape', 'GFLOP', 'fwd us', 'TF', 'dgrad us', 'TF', 'wgrad us', 'TF'))
ONLY = set(os.environ.get('ONLY', '').split(',')) - {''}
for cin, cout, k, stv, pad, H, cnt in SHAPES:
    if ONLY and ('%d-%d-%d' % (cin, cout, k)) not in ONLY:
        continue
    g = P.ConvGeom(cin, cout, k, stv, pad)
    fd = P.fwd_desc(g, N, H, H)
    dds, skipped = P.dgrad_plan(g, N, H, H)
    wd = P.wgrad_desc(g, N, H, H)
    packer = WeightPacker()
    for d in [fd] + dds:
        packer.add(0, cout, k, k, cin, d.pack)
    w = torch.randn(cout * k * k * cin, device=DEV) * 0.05
    packer.build(DEV, dtype).run(w)
    x = torch.randn(N, H, H, cin, device=DEV).to(dtype)
    y = torch.empty(N, fd.OP, fd.OQ, cout, device=DEV, dtype=dtype)
    dy = torch.randn(N, fd.OP, fd.OQ, cout, device=DEV).to(dtype)
    dx = torch.zeros(N, H, H, cin, device=DEV, dtype=dtype)
    dw = torch.zeros(cout, k * k * cin, device=DEV)
    fl = 2.0 * N * fd.OP * fd.OQ * cout * k * k * cin
    t_f = timeit(lambda: ops.conv_igemm(fd, x, packer.view(fd.pack, cout), y))
    st, _tiles = ops.conv_stats_buffer(fd, DEV)
    t_fs = timeit(lambda: ops.conv_igemm(fd, x, packer.view(fd.pack, cout), y, stats=st)) if dtype == torch.bfloat16 else 0
    def dg():
        for d in dds: ops.conv_igemm(d, dy, packer.view(d.pack, cin), dx)
    t_d = timeit(dg)
    t_w = timeit(lambda: ops.conv_wgrad(wd, x, dy.view(-1, cout), dw))
    for nm, t in (('fwd', t_f), ('dgrad', t_d), ('wgrad', t_w)):
        tot[nm][0] += t * cnt; tot[nm][1] += fl * cnt
    tot.setdefault('fwd+stats', [0, 0]); tot['fwd+stats'][0] += t_fs * cnt; tot['fwd+stats'][1] += fl * cnt
    print('%4d->%4d k%d s%d @%3d x%d       %8.2f | %8.1f %7.1f | %8.1f %7.1f | %8.1f %7.1f | stats %8.1f tiles=%d' % (
        cin, cout, k, stv, H, cnt, fl / 1e9, t_f * 1e3, fl / t_f / 1e9, t_d * 1e3, fl / t_d / 1e9, t_w * 1e3, fl / t_w / 1e9, t_fs * 1e3, _tiles))
for nm in tot:
    print('%s total: %.2f ms per pass, %.1f TF avg' % (nm, tot[nm][0], tot[nm][1] / tot[nm][0] / 1e9))
