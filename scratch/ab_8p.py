"""igemm_8p_kernel (256 x 256 tiles, 8-phase) vs igemm_ring_kernel: bit-exactness (outputs, fused statistics,
fused BatchNorm-backward slabs), a race screen (repeated launches must agree bit for bit) and timing on the
ViT Linear shapes, the R50 layers with >= 256 output channels and square GEMMs."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from passl_amd.hip import ops, plan as P, lib as L
from passl_amd.hip.packer import WeightPacker

lib = L.load(); DEV = 'cuda'; dtype = torch.bfloat16
QUICK = '--quick' in sys.argv


def mode(m):
    lib.passl_hip_set_option(b'igemm_8p', m)


def run(fn, iters=20):
    for _ in range(3): fn()
    s = torch.cuda.Event(True); e = torch.cuda.Event(True); s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / iters * 1e3


def exact(name, fn, outs, reps=6):
    """outs: tensors the launch writes.  ring (mode 0) vs forced 8p (mode 2), then 8p repeated."""
    mode(0)
    for o in outs: o.zero_()
    fn(); ref = [o.clone() for o in outs]; kr = lib.passl_hip_last_igemm_kernel()
    mode(2)
    bad = 0; k8 = None
    for r in range(reps):
        for o in outs: o.fill_(float('nan')) if o.dtype.is_floating_point else o.zero_()
        fn(); k8 = lib.passl_hip_last_igemm_kernel()
        for o, a in zip(outs, ref):
            same = torch.equal(o.view(torch.int16 if o.dtype == torch.bfloat16 else torch.int32),
                               a.view(torch.int16 if a.dtype == torch.bfloat16 else torch.int32))
            if not same:
                bad += 1
                if bad <= 2:
                    d = (o.float() - a.float()).abs()
                    print('   MISMATCH rep %d: max %.3e nnz %d / %d nan %d' % (r, d.nan_to_num(1e9).max().item(), int((d != 0).sum()), d.numel(), int(torch.isnan(o.float()).sum())))
    print('%-58s kernels %d vs %d  %s' % (name, kr, k8, 'EXACT x%d' % reps if bad == 0 else 'FAIL (%d)' % bad), flush=True)
    return bad == 0


def conv_case(cin, cout, k, st, pad, H, N, do_epi=True):
    g = P.ConvGeom(cin, cout, k, st, pad); fd = P.fwd_desc(g, N, H, H); dds, sk = P.dgrad_plan(g, N, H, H)
    packer = WeightPacker()
    for d in [fd] + dds: packer.add(0, cout, k, k, cin, d.pack)
    w = torch.randn(cout * k * k * cin, device=DEV) * 0.05
    packer.build(DEV, dtype).run(w)
    x = torch.randn(N, H, H, cin, device=DEV).to(dtype); y = torch.zeros(N, fd.OP, fd.OQ, cout, device=DEV, dtype=dtype)
    dy = torch.randn(N, fd.OP, fd.OQ, cout, device=DEV).to(dtype); dx = torch.zeros(N, H, H, cin, device=DEV, dtype=dtype)
    tag = '%d->%d k%d s%d @%d N%d' % (cin, cout, k, st, H, N)
    ok = exact('fwd   ' + tag, lambda: ops.conv_igemm(fd, x, packer.view(fd.pack, cout), y), [y])
    # torch reference (fp32 accumulate of the bf16 operands)
    wl = packer.view(fd.pack, cout).float().view(cout, k, k, cin).permute(0, 3, 1, 2)
    ref = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), wl, stride=st, padding=pad).permute(0, 2, 3, 1)
    err = (y.float() - ref).abs().max().item() / ref.abs().max().item()
    print('      vs torch fp32 conv: rel max err %.2e' % err)
    ok &= err < 1e-2
    def dg():
        for d in dds: ops.conv_igemm(d, dy, packer.view(d.pack, cin), dx)
    ok &= exact('dgrad ' + tag, dg, [dx])
    if do_epi:
        sc = torch.rand(cout, device=DEV) + 0.5; sh = torch.randn(cout, device=DEV); res = torch.randn_like(y)
        ok &= exact('fwd+affine+res+relu', lambda: ops.conv_igemm(fd, x, packer.view(fd.pack, cout), y, scale=sc, shift=sh, residual=res, relu=True), [y])
        slab, tiles = ops.conv_stats_buffer(fd, DEV)
        ok &= exact('fwd+stats', lambda: ops.conv_igemm(fd, x, packer.view(fd.pack, cout), y, stats=slab), [y, slab[:tiles * cout * 3]])
        if len(dds) == 1 and not sk:
            d = dds[0]
            t = ops.conv_tiles(d)
            part = torch.zeros(ops.bn_partial_floats(t, cin, False), device=DEV)
            yb = torch.randn(N, H, H, cin, device=DEV).to(dtype)
            mean = torch.randn(cin, device=DEV) * 0.1; inv = torch.rand(cin, device=DEV) + 0.5
            bsc = torch.rand(cin, device=DEV) + 0.5; bsh = torch.randn(cin, device=DEV) * 0.3
            bnb = dict(y=yb, mask=None, mean=mean, invstd=inv, scale=bsc, shift=bsh, relu=2, partial=part, tile_off=0)
            ok &= exact('dgrad+bnb(relu2)', lambda: ops.conv_igemm(d, dy, packer.view(d.pack, cin), dx, bnb=bnb), [dx, part[:t * cin * 2]])
    return ok


def linear_case(M, cin, cout):
    g = P.ConvGeom(cin, cout, 1, 1, 0); fd = P.fwd_desc(g, M, 1, 1)
    packer = WeightPacker(); packer.add(0, cout, 1, 1, cin, fd.pack)
    packer.build(DEV, dtype).run(torch.randn(cout * cin, device=DEV) * 0.05)
    x = torch.randn(M, cin, device=DEV).to(dtype); y = torch.zeros(M, cout, device=DEV, dtype=dtype)
    b = torch.randn(cout, device=DEV); res = torch.randn(M, cout, device=DEV).to(dtype)
    ok = exact('linear M%d %d->%d' % (M, cin, cout), lambda: ops.conv_igemm(fd, x, packer.view(fd.pack, cout), y, shift=b, residual=res), [y])
    ref = x.float() @ packer.view(fd.pack, cout).float().t() + b + res.float()
    err = (y.float() - ref).abs().max().item() / ref.abs().max().item()
    print('      vs torch fp32: rel max err %.2e' % err)
    return ok and err < 1e-2


all_ok = True
print('== exactness / race screen (reference = the ring kernel wherever it applies)')
lib.passl_hip_set_option(b'igemm_ring_min_nk', 1)
for args in [(256, 256, 3, 1, 1, 14, 8), (512, 512, 3, 1, 1, 7, 32), (512, 512, 3, 2, 1, 14, 8), (1024, 256, 1, 1, 0, 14, 16),
             (256, 1024, 1, 1, 0, 14, 5), (2048, 512, 1, 1, 0, 7, 11), (512, 2048, 1, 1, 0, 7, 32),
             (128, 320, 3, 1, 1, 9, 3)]:
    all_ok &= conv_case(*args)
for M, cin, cout in [(12800, 768, 2304), (12800, 3072, 768), (50432, 512, 1536), (1000, 768, 768), (257, 64, 264), (4096, 4096, 4096),
                     (70000, 192, 520), (66000, 128, 256), (3000, 320, 8)]:
    all_ok &= linear_case(M, cin, cout)
print('ALL EXACT' if all_ok else 'SOME FAILED', flush=True)
lib.passl_hip_set_option(b'igemm_ring_min_nk', 8)

print('== timing (us, TFLOP/s): ring | 8p staged (forced) | 8p persistent (forced) | auto (cost model)')
SH = [(12800, 768, 2304, 'mae-enc qkv'), (12800, 768, 768, 'mae-enc proj'), (12800, 768, 3072, 'mae-enc fc1'),
      (12800, 3072, 768, 'mae-enc fc2'), (50432, 512, 1536, 'mae-dec qkv'), (50432, 512, 512, 'mae-dec proj'),
      (50432, 512, 2048, 'mae-dec fc1'), (50432, 2048, 512, 'mae-dec fc2'),
      (50432, 768, 2304, 'clip16 qkv'), (50432, 768, 768, 'clip16 proj'), (50432, 768, 3072, 'clip16 fc1'), (50432, 3072, 768, 'clip16 fc2'),
      (19712, 512, 1536, 'clip-txt qkv'), (19712, 512, 2048, 'clip-txt fc1'), (19712, 2048, 512, 'clip-txt fc2'),
      (4096, 4096, 4096, 'square 4k'), (8192, 8192, 8192, 'square 8k')]
if QUICK: SH = SH[:4]
for M, cin, cout, label in SH:
    g = P.ConvGeom(cin, cout, 1, 1, 0); fd = P.fwd_desc(g, M, 1, 1)
    packer = WeightPacker(); packer.add(0, cout, 1, 1, cin, fd.pack)
    packer.build(DEV, dtype).run(torch.randn(cout * cin, device=DEV) * 0.05)
    x = torch.randn(M, cin, device=DEV).to(dtype); y = torch.empty(M, cout, device=DEV, dtype=dtype)
    f = lambda: ops.conv_igemm(fd, x, packer.view(fd.pack, cout), y)
    fl = 2.0 * M * cin * cout
    out = []
    for m, dr in ((0, 1), (2, 0), (2, 1), (1, 1)):
        mode(m); lib.passl_hip_set_option(b'igemm_8p_direct', dr)
        t = run(f); out.append('%7.1f %5.0f k%d' % (t, fl / t / 1e6, lib.passl_hip_last_igemm_kernel()))
    print('%-14s M=%5d %4d->%4d %7.1f GF | ' % (label, M, cin, cout, fl / 1e9) + ' | '.join(out), flush=True)

R50 = [(256, 256, 3, 2, 1, 28), (256, 256, 3, 1, 1, 14), (1024, 256, 1, 1, 0, 14), (256, 1024, 1, 1, 0, 14), (512, 1024, 1, 2, 0, 28),
       (1024, 512, 1, 1, 0, 14), (512, 512, 3, 2, 1, 14), (512, 512, 3, 1, 1, 7), (512, 2048, 1, 1, 0, 7), (2048, 512, 1, 1, 0, 7),
       (1024, 2048, 1, 2, 0, 14), (512, 256, 1, 1, 0, 28), (256, 512, 1, 2, 0, 56), (128, 512, 1, 1, 0, 28), (64, 256, 1, 1, 0, 56)]
if QUICK: R50 = R50[:3]
N = 256
for cin, cout, k, st, pad, H in R50:
    g = P.ConvGeom(cin, cout, k, st, pad); fd = P.fwd_desc(g, N, H, H); dds, sk = P.dgrad_plan(g, N, H, H)
    packer = WeightPacker()
    for d in [fd] + dds: packer.add(0, cout, k, k, cin, d.pack)
    packer.build(DEV, dtype).run(torch.randn(cout * k * k * cin, device=DEV) * 0.05)
    x = torch.randn(N, H, H, cin, device=DEV).to(dtype); y = torch.empty(N, fd.OP, fd.OQ, cout, device=DEV, dtype=dtype)
    dy = torch.randn(N, fd.OP, fd.OQ, cout, device=DEV).to(dtype); dx = torch.zeros(N, H, H, cin, device=DEV, dtype=dtype)
    slab, tiles = ops.conv_stats_buffer(fd, DEV)
    fl = 2.0 * N * fd.OP * fd.OQ * cout * k * k * cin
    f = lambda: ops.conv_igemm(fd, x, packer.view(fd.pack, cout), y, stats=slab)
    def dg():
        for d in dds: ops.conv_igemm(d, dy, packer.view(d.pack, cin), dx)
    row = []
    for nm, fn in (('fwd+stats', f), ('dgrad', dg)):
        o = []
        for m, dr in ((0, 1), (2, 0), (2, 1), (1, 1)):
            if nm == 'fwd+stats' and (m, dr) == (2, 1): continue      # statistics keep the staged epilogue
            mode(m); lib.passl_hip_set_option(b'igemm_8p_direct', dr)
            t = run(fn); o.append('%6.1f %4.0f k%d' % (t, fl / t / 1e6, lib.passl_hip_last_igemm_kernel()))
        row.append(nm + ' ' + ' | '.join(o))
    print('%4d->%4d k%d s%d @%2d  %5.1f GF  ' % (cin, cout, k, st, H, fl / 1e9) + '  ||  '.join(row), flush=True)
mode(1)
