"""A/B on ViT Linear shapes (bf16): fwd igemm_kernel vs ring<256> vs ring<128>; wgrad over split counts."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from passl_amd.hip import ops, plan as P, lib as L
from passl_amd.hip.packer import WeightPacker
lib = L.load()
DEV = 'cuda'; dtype = torch.bfloat16
# (M, in, out, label)
SHAPES = [(12800, 768, 2304, 'mae-enc qkv'), (12800, 768, 768, 'mae-enc proj'), (12800, 768, 3072, 'mae-enc fc1'),
          (12800, 3072, 768, 'mae-enc fc2'), (50432, 512, 1536, 'mae-dec qkv'), (50432, 512, 512, 'mae-dec proj'),
          (50432, 512, 2048, 'mae-dec fc1'), (50432, 2048, 512, 'mae-dec fc2'),
          (6400, 768, 2304, 'clip-img qkv'), (6400, 768, 3072, 'clip-img fc1'), (6400, 3072, 768, 'clip-img fc2'),
          (9856, 512, 1536, 'clip-txt qkv'), (9856, 512, 2048, 'clip-txt fc1'), (9856, 2048, 512, 'clip-txt fc2')]
def run(fn, iters=20):
    for _ in range(3): fn()
    s = torch.cuda.Event(True); e = torch.cuda.Event(True); s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / iters * 1e3
for M, cin, cout, label in SHAPES:
    g = P.ConvGeom(cin, cout, 1, 1, 0); fd = P.fwd_desc(g, M, 1, 1); wd = P.wgrad_desc(g, M, 1, 1)
    packer = WeightPacker(); packer.add(0, cout, 1, 1, cin, fd.pack)
    packer.build(DEV, dtype).run(torch.randn(cout * cin, device=DEV) * 0.05)
    x = torch.randn(M, cin, device=DEV).to(dtype); y = torch.empty(M, cout, device=DEV, dtype=dtype)
    dy = torch.randn(M, cout, device=DEV).to(dtype); dw = torch.zeros(cout, cin, device=DEV)
    fl = 2.0 * M * cin * cout
    f = lambda: ops.conv_igemm(fd, x, packer.view(fd.pack, cout), y)
    out = []
    lib.passl_hip_set_option(b'igemm_ring', 0); out.append(('old', run(f)))
    lib.passl_hip_set_option(b'igemm_ring', 1); lib.passl_hip_set_option(b'igemm_ring_min_tiles', 1)
    lib.passl_hip_set_option(b'igemm_ring_min_nk', 1)
    lib.passl_hip_set_option(b'igemm_ring_bm', 256); out.append(('r256', run(f)))
    lib.passl_hip_set_option(b'igemm_ring_bm', 128); out.append(('r128', run(f)))
    tiles = ((cout + 127) // 128) * ((cin + 127) // 128)
    dflt = P.wgrad_splits(M, cout, cin, 64)
    ws = []
    for tile, nm in ((4, 'T128x128'), (2, 'T64x128'), (3, 'T128x64'), (1, 'T64x64')):
        lib.passl_hip_set_option(b'wgrad_tile', tile)
        best = min((run(lambda: ops.conv_wgrad(wd, x, dy, dw, splits=sp)), sp) for sp in (1, 2, 3, 4, 6, 8, 12, 16) if sp * 64 <= M)
        ws.append((nm + ':s%d' % best[1], best[0]))
    lib.passl_hip_set_option(b'wgrad_tile', 0)
    print('%-14s M=%5d %4d->%4d %6.1f GF | fwd ' % (label, M, cin, cout, fl / 1e9) +
          ' '.join('%s %6.1f us (%4.0f TF)' % (n, t, fl / t / 1e6) for n, t in out) +
          ' | wgrad tiles=%3d dflt=%2d: ' % (tiles, dflt) + ' '.join('%s %.0f(%.0fTF)' % (sp, t, fl / t / 1e6) for sp, t in ws))
