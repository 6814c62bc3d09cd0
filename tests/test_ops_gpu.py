"""Per-kernel parity tests on a real MI355X: every HIP entry point of libpassl_hip.so, called
through the C ABI (ctypes), against a plain PyTorch fp32 reference of the same op.

Tolerances: fp32 kernels 1e-4 relative-to-max (exact-fp32 MFMA, different summation order);
bf16 kernels compare against the fp32 reference evaluated on bf16-rounded inputs, 2e-2
relative-to-max on outputs that are themselves rounded to bf16.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import emu                                    # noqa: E402
from passl_amd.hip import ops, plan as P      # noqa: E402
from passl_amd.hip.packer import WeightPacker  # noqa: E402

DEV = 'cuda'
DTYPES = [torch.float32, torch.bfloat16]


def relmax(a, b):
    a = a.double().cpu()
    b = b.double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


def tol(dtype):
    return 1e-4 if dtype == torch.float32 else 2e-2


def rnd(t, dtype):
    """round-trip through the compute dtype so the fp32 reference sees the same inputs"""
    return t.to(dtype).float()


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


# ---------------------------------------------------------------- flat kernels
@pytest.mark.parametrize('n', [1024, 4099, 3_000_003])
def test_ema_sgd_cast(n):
    g = torch.Generator().manual_seed(n)
    k = torch.randn(n, generator=g)
    q = torch.randn(n, generator=g)
    kd, qd = k.to(DEV), q.to(DEV)
    klp = torch.empty(n, dtype=torch.bfloat16, device=DEV)
    ops.ema_update(kd, qd, 0.999, klp)
    ref = k * 0.999 + q * (1 - 0.999)
    assert (kd.cpu() - ref).abs().max() < 1e-6
    # the bf16 copy is the RNE rounding of the kernel's own fp32 result (GPU fma vs CPU mul+add
    # can differ by 1 fp32 ulp, which may flip a bf16 tie)
    assert (klp.float() - kd.to(torch.bfloat16).float()).abs().max().item() == 0
    # momentum sgd, two steps
    p = torch.randn(n, generator=g); gr = torch.randn(n, generator=g); v = torch.zeros(n)
    pd, gd, vd = p.to(DEV), gr.to(DEV), v.to(DEV)
    for lr in (0.015, 0.01):
        ops.momentum_sgd(pd, gd, vd, lr, 0.9, 1e-4, 1.0)
        gg = gr + 1e-4 * p
        v = 0.9 * v + gg
        p = p - lr * v
    assert (pd.cpu() - p).abs().max() < 1e-5
    assert (vd.cpu() - v).abs().max() < 1e-5
    dst = torch.empty(n, dtype=torch.bfloat16, device=DEV)
    ops.cast_bf16(qd, dst)
    assert (dst.float().cpu() - q.to(torch.bfloat16).float()).abs().max() == 0


def test_nchw_to_nhwc_pad():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(3, 3, 20, 18, generator=g)
    for dtype in DTYPES:
        Hp, Wp = P.stem_padded_hw(20, 18)
        y = ops.nchw_to_nhwc_pad(x.to(DEV), 3, Wp, 4, dtype)
        ref = torch.zeros(3, Hp, Wp, 4)
        ref[:, 3:23, 3:21, :3] = nhwc(x)
        assert (y.float().cpu() - rnd(ref, dtype)).abs().max() == 0


# ---------------------------------------------------------------- convolution
GEOMS = [
    (P.ConvGeom(64, 64, 1, 1, 0), (4, 28, 28)),
    (P.ConvGeom(64, 64, 3, 1, 1), (4, 28, 28)),
    (P.ConvGeom(64, 256, 1, 1, 0), (2, 56, 56)),
    (P.ConvGeom(128, 128, 3, 2, 1), (3, 28, 28)),
    (P.ConvGeom(256, 512, 1, 2, 0), (3, 14, 14)),
    (P.ConvGeom(512, 128, 1, 1, 0), (2, 7, 7)),       # M = 98 (ragged M tile)
    (P.ConvGeom(256, 256, 3, 1, 1), (2, 14, 14)),
    (P.ConvGeom(128, 64, 3, 1, 1), (1, 9, 11)),       # odd spatial
    (P.ConvGeom(72, 136, 3, 2, 1), (2, 10, 10)),      # C not a multiple of the K tile (generic path)
]


def _run_plan(descs, a, packs_buf, packer, y, **kw):
    for d in descs:
        rows = d.NCOLS
        ops.conv_igemm(d, a, packer.view(d.pack, rows), y, **kw)
    return y


@pytest.fixture
def force_ring_kernel():
    """Route every eligible bf16 conv through the LDS-DMA ring kernel (normally only reductions
    of >= 8 K-tiles use it), in both tile configurations."""
    from passl_amd.hip import lib as L
    lib = L.load()
    assert lib.passl_hip_set_option(b'igemm_ring_min_nk', 1) == 0

    def select(bm):
        assert lib.passl_hip_set_option(b'igemm_ring_bm', bm) == 0
    yield select
    assert lib.passl_hip_set_option(b'igemm_ring_min_nk', 8) == 0
    assert lib.passl_hip_set_option(b'igemm_ring_bm', 128) == 0
    assert lib.passl_hip_set_option(b'bogus', 1) != 0 and lib.passl_hip_set_option(b'igemm_ring_bm', 7) != 0


@pytest.mark.parametrize('bm', [128, 256])
@pytest.mark.parametrize('geom,nhw', [g for g in GEOMS if g[0].cin % 64 == 0])
def test_conv_ring_kernel(geom, nhw, bm, force_ring_kernel):
    force_ring_kernel(bm)
    test_conv_fwd_dgrad_wgrad(geom, nhw, torch.bfloat16)


@pytest.mark.parametrize('bm', [128, 256])
def test_conv_ring_kernel_epilogue(bm, force_ring_kernel):
    force_ring_kernel(bm)
    test_conv_epilogue_affine_residual_relu_f32out(torch.bfloat16)


@pytest.fixture
def kernel_8p():
    """Select the 256 x 256-tile 8-phase kernel: mode 0 never / 1 cost model / 2 whenever it applies; `direct`
    0 keeps the staged epilogue even where the persistent form would run."""
    from passl_amd.hip import lib as L
    lib = L.load()

    def select(mode, direct=1):
        assert lib.passl_hip_set_option(b'igemm_8p', mode) == 0
        assert lib.passl_hip_set_option(b'igemm_8p_direct', direct) == 0
        return lib
    yield select
    assert lib.passl_hip_set_option(b'igemm_8p', 1) == 0
    assert lib.passl_hip_set_option(b'igemm_8p_direct', 1) == 0
    assert lib.passl_hip_set_option(b'igemm_ring_min_nk', 8) == 0
    assert lib.passl_hip_set_option(b'igemm_8p', 3) != 0


@pytest.mark.parametrize('geom,nhw', [g for g in GEOMS if g[0].cin % 64 == 0])
def test_conv_8p_kernel(geom, nhw, kernel_8p):
    """every GEOMS shape inside the 8-phase kernel's envelope, forced through it (persistent form)"""
    kernel_8p(2)
    test_conv_fwd_dgrad_wgrad(geom, nhw, torch.bfloat16)


@pytest.mark.parametrize('direct', [0, 1])
def test_conv_8p_kernel_epilogue(direct, kernel_8p):
    kernel_8p(2, direct)
    test_conv_epilogue_affine_residual_relu_f32out(torch.bfloat16)


def _bits(t):
    return t.view(torch.int16 if t.dtype == torch.bfloat16 else torch.int32)


# (M, K, N): several tiles per workgroup of the persistent form (> 256 tiles), odd / minimal K-tile counts,
# ragged last row tile and last column tile, one ViT shape
LINEAR_8P = [(70000, 192, 520), (66000, 128, 256), (12800, 768, 2304), (1000, 3072, 768), (257, 64, 264)]


@pytest.mark.parametrize('M,K,N', LINEAR_8P)
def test_8p_kernel_bit_identical_to_ring_kernel(M, K, N, kernel_8p):
    """The 8-phase kernel (staged and persistent forms) accumulates in the same order and applies the same epilogue
    arithmetic as the ring kernel: bias + residual + ReLU outputs are bit-identical, and repeated launches agree
    with each other (race screen of the DMA / barrier schedule)."""
    lib = kernel_8p(0)
    assert lib.passl_hip_set_option(b'igemm_ring_min_nk', 1) == 0
    gen = torch.Generator().manual_seed(M + K)
    g = P.ConvGeom(K, N, 1, 1, 0)
    fd = P.fwd_desc(g, M, 1, 1)
    packer = WeightPacker()
    packer.add(0, N, 1, 1, K, fd.pack)
    packer.build(DEV, torch.bfloat16).run((torch.randn(N * K, generator=gen) * 0.05).to(DEV))
    x = torch.randn(M, K, generator=gen).to(DEV).to(torch.bfloat16)
    b = torch.randn(N, generator=gen).to(DEV)
    res = torch.randn(M, N, generator=gen).to(DEV).to(torch.bfloat16)
    y = torch.zeros(M, N, device=DEV, dtype=torch.bfloat16)

    def launch(**kw):
        y.fill_(float('nan'))
        ops.conv_igemm(fd, x, packer.view(fd.pack, N), y, **kw)
        return y.clone(), lib.passl_hip_last_igemm_kernel()
    for kw in (dict(), dict(shift=b, residual=res, relu=True)):
        kernel_8p(0)
        ref, k_ref = launch(**kw)
        assert k_ref in (0, 1)
        want = x.float() @ packer.view(fd.pack, N).float().t()
        if kw:
            want = torch.relu(want + b + res.float())
        assert relmax(ref, want) < 2e-2
        for direct in (0, 1):
            kernel_8p(2, direct)
            for rep in range(4):
                out, k_out = launch(**kw)
                assert k_out == 3
                assert torch.equal(_bits(out), _bits(ref)), (direct, rep, kw.keys())


def test_8p_kernel_fused_statistics_bit_identical(kernel_8p):
    """conv + fused BatchNorm statistics and data-gradient + fused BatchNorm-backward slabs through the staged form
    of the 8-phase kernel: outputs and slabs equal the ring kernel's bit for bit (128 columns per ring tile)."""
    lib = kernel_8p(0)
    assert lib.passl_hip_set_option(b'igemm_ring_min_nk', 1) == 0
    gen = torch.Generator().manual_seed(5)
    for cin, cout, k, st, H, N in ((256, 256, 3, 1, 14, 8), (128, 320, 3, 1, 9, 3), (512, 256, 1, 1, 7, 11)):
        g = P.ConvGeom(cin, cout, k, st, k // 2)
        fd = P.fwd_desc(g, N, H, H)
        dds, skipped = P.dgrad_plan(g, N, H, H)
        assert len(dds) == 1 and not skipped
        packer = WeightPacker()
        for d in [fd] + dds:
            packer.add(0, cout, k, k, cin, d.pack)
        packer.build(DEV, torch.bfloat16).run((torch.randn(cout * k * k * cin, generator=gen) * 0.05).to(DEV))
        x = torch.randn(N, H, H, cin, generator=gen).to(DEV).to(torch.bfloat16)
        y = torch.zeros(N, fd.OP, fd.OQ, cout, device=DEV, dtype=torch.bfloat16)
        dy = torch.randn(N, fd.OP, fd.OQ, cout, generator=gen).to(DEV).to(torch.bfloat16)
        dx = torch.zeros(N, H, H, cin, device=DEV, dtype=torch.bfloat16)
        slab, tiles = ops.conv_stats_buffer(fd, DEV)
        d = dds[0]
        td = ops.conv_tiles(d)
        part = torch.zeros(ops.bn_partial_floats(td, cin, False), device=DEV)
        yb = torch.randn(N, H, H, cin, generator=gen).to(DEV).to(torch.bfloat16)
        bnb = dict(y=yb, mask=None, mean=torch.randn(cin, generator=gen).to(DEV) * 0.1,
                   invstd=torch.rand(cin, generator=gen).to(DEV) + 0.5, scale=torch.rand(cin, generator=gen).to(DEV) + 0.5,
                   shift=torch.randn(cin, generator=gen).to(DEV) * 0.3, relu=2, partial=part, tile_off=0)
        got = {}
        for mode in (0, 2):
            kernel_8p(mode)
            slab.fill_(float('nan')); part.fill_(float('nan'))
            ops.conv_igemm(fd, x, packer.view(fd.pack, cout), y, stats=slab)
            kf = lib.passl_hip_last_igemm_kernel()
            ops.conv_igemm(d, dy, packer.view(d.pack, cin), dx, bnb=bnb)
            kd = lib.passl_hip_last_igemm_kernel()
            assert (kf, kd) == ((1, 1) if mode == 0 else (3, 3))
            got[mode] = [t.clone() for t in (y, slab[:tiles * cout * 3], dx, part[:td * cin * 2])]
        for a, b in zip(got[0], got[2]):
            assert not torch.isnan(b.float()).any()
            assert torch.equal(_bits(a), _bits(b))


def test_8p_cost_model_dispatch(kernel_8p):
    """mode 1: wide and deep launches whose 256 x 256 tiles fill the chip go to the 8-phase kernel, short reductions
    and launches with too few tiles do not."""
    lib = kernel_8p(1)
    for (M, K, N), want in (((50432, 768, 2304), 3), ((50176, 1024, 256), 3), ((50176, 256, 1024), 0),
                            ((12544, 2048, 512), 1), ((200704, 512, 128), 1)):
        g = P.ConvGeom(K, N, 1, 1, 0)
        fd = P.fwd_desc(g, M, 1, 1)
        w = torch.zeros(N, K, device=DEV, dtype=torch.bfloat16)
        x = torch.zeros(M, K, device=DEV, dtype=torch.bfloat16)
        y = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
        ops.conv_igemm(fd, x, w, y)
        assert lib.passl_hip_last_igemm_kernel() == want, (M, K, N)


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('geom,nhw', GEOMS)
def test_conv_fwd_dgrad_wgrad(geom, nhw, dtype):
    N, H, W = nhw
    gen = torch.Generator().manual_seed(7)
    x = rnd(torch.randn(N, geom.cin, H, W, generator=gen), dtype).requires_grad_(True)
    w = rnd(torch.randn(geom.cout, geom.cin, geom.k, geom.k, generator=gen) * 0.1, dtype).requires_grad_(True)
    y = F.conv2d(x.double(), w.double(), None, geom.stride, geom.pad)
    dy = rnd(torch.randn(y.shape, generator=gen), dtype)
    y.backward(dy.double())
    # master weights, physically KRSC fp32
    w_krsc = w.detach().permute(0, 2, 3, 1).contiguous().to(DEV)
    fd = P.fwd_desc(geom, N, H, W)
    dds, skipped = P.dgrad_plan(geom, N, H, W)
    packer = WeightPacker()
    for d in [fd] + dds:
        packer.add(0, geom.cout, geom.k, geom.k, geom.cin, d.pack)
    packer.build(DEV, dtype).run(w_krsc.view(-1))
    # pack kernel vs emulator
    for d in [fd] + dds:
        got = packer.buffer[d.pack.dst_off:d.pack.dst_off + d.pack.size].float().cpu()
        ref = emu.emu_pack(d.pack, w.detach().permute(0, 2, 3, 1).contiguous()).reshape(-1)
        assert (got - rnd(ref, dtype)).abs().max() == 0
    xa = nhwc(x.detach()).to(DEV).to(dtype)
    yo = torch.empty(N, fd.OP, fd.OQ, geom.cout, dtype=dtype, device=DEV)
    ops.conv_igemm(fd, xa, packer.view(fd.pack, geom.cout), yo)
    assert relmax(yo.float(), nhwc(y.detach())) < tol(dtype)
    if dtype == torch.bfloat16:
        # fused BatchNorm statistics of the conv epilogue: per 128-row tile t the SHIFTED sums
        # (sum (v - s), sum (v - s)^2) of the STORED values, s = the tile's first row; plain stores into
        # a slab (no atomics, no zeroing) -> bit-identical run to run; same output as without statistics
        st, tiles = ops.conv_stats_buffer(fd, DEV)
        st.fill_(float('nan'))
        yo2 = torch.empty_like(yo)
        ops.conv_igemm(fd, xa, packer.view(fd.pack, geom.cout), yo2, stats=st)
        assert torch.equal(yo2, yo)
        assert not torch.isnan(st[:tiles * geom.cout * 3]).any()
        y64 = yo.double().cpu().reshape(-1, geom.cout)
        M = y64.shape[0]
        sums = st[:tiles * geom.cout * 2].view(tiles, geom.cout, 2).double().cpu()
        shifts = st[tiles * geom.cout * 2:tiles * geom.cout * 3].view(tiles, geom.cout).double().cpu()
        for t in range(tiles):
            rows = y64[t * 128:min((t + 1) * 128, M)]
            assert torch.equal(shifts[t], rows[0])
            d = rows - rows[0]
            assert (sums[t, :, 0] - d.sum(0)).abs().max() <= 1e-5 * max(1.0, float(d.abs().sum(0).max()))
            assert (sums[t, :, 1] - (d * d).sum(0)).abs().max() <= 1e-5 * max(1.0, float((d * d).sum(0).max()))
        st_b = torch.empty_like(st)
        ops.conv_igemm(fd, xa, packer.view(fd.pack, geom.cout), yo2, stats=st_b)
        assert torch.equal(st_b[:tiles * geom.cout * 3], st[:tiles * geom.cout * 3])     # reproducible
        # ... and the finalize kernel turns the slab into the statistics of the stored values
        ga, be = torch.ones(geom.cout, device=DEV), torch.zeros(geom.cout, device=DEV)
        rm, rv = torch.zeros(geom.cout, device=DEV), torch.ones(geom.cout, device=DEV)
        _z, st4, _m = ops.bn_train_fwd(yo, ga, be, rm, rv, relu=False, partial=(st, tiles))
        mu, var = y64.mean(0), y64.var(0, unbiased=False)
        assert (st4[0].double().cpu() - mu).abs().max() < 1e-5 * max(1.0, float(mu.abs().max()))
        assert relmax(st4[1], 1 / torch.sqrt(var + 1e-5)) < 1e-5
    # dgrad
    dya = nhwc(dy).to(DEV).to(dtype)
    dx = torch.full((N, H, W, geom.cin), float('nan'), dtype=dtype, device=DEV)
    if skipped:
        dx.zero_()
    for d in dds:
        ops.conv_igemm(d, dya, packer.view(d.pack, geom.cin), dx)
    assert not torch.isnan(dx.float()).any()
    assert relmax(dx.float(), nhwc(x.grad)) < tol(dtype)
    if len(dds) == 1 and not skipped:
        # residual-fork gradient added in the data-gradient epilogue (nn.GradSlot path)
        extra = rnd(torch.randn(N, H, W, geom.cin, generator=gen), dtype)
        dx2 = torch.empty_like(dx)
        ops.conv_igemm(dds[0], dya, packer.view(dds[0].pack, geom.cin), dx2,
                       residual=extra.to(DEV).to(dtype))
        assert relmax(dx2.float(), nhwc(x.grad) + extra.double()) < tol(dtype)
    # wgrad (fp32 accumulate, several split counts)
    for splits in (None, 1, 3):
        dw = torch.zeros(geom.cout, geom.k * geom.k * geom.cin, dtype=torch.float32, device=DEV)
        ops.conv_wgrad(P.wgrad_desc(geom, N, H, W), xa, dya.view(-1, geom.cout), dw, splits)
        ref = w.grad.permute(0, 2, 3, 1).reshape(geom.cout, -1)
        assert relmax(dw, ref) < (1e-4 if dtype == torch.float32 else 2e-3), splits
        # split-M partial tiles are summed in slice order (workspace slabs): reproducible bit for bit,
        # and the launch ACCUMULATES into dw
        dw2 = torch.zeros_like(dw)
        ops.conv_wgrad(P.wgrad_desc(geom, N, H, W), xa, dya.view(-1, geom.cout), dw2, splits)
        assert torch.equal(dw2, dw), splits
        ops.conv_wgrad(P.wgrad_desc(geom, N, H, W), xa, dya.view(-1, geom.cout), dw2, splits)
        assert relmax(dw2, 2 * ref) < (1e-4 if dtype == torch.float32 else 2e-3), splits


@pytest.mark.parametrize('dtype', DTYPES)
def test_conv_epilogue_affine_residual_relu_f32out(dtype):
    geom = P.ConvGeom(64, 128, 1, 1, 0)
    N, H, W = 2, 12, 12
    gen = torch.Generator().manual_seed(3)
    x = rnd(torch.randn(N, 64, H, W, generator=gen), dtype)
    w = rnd(torch.randn(128, 64, 1, 1, generator=gen) * 0.1, dtype)
    scale = torch.rand(128, generator=gen) + 0.5
    shift = torch.randn(128, generator=gen)
    res = rnd(torch.randn(N, 128, H, W, generator=gen), dtype)
    ref = F.relu(F.conv2d(x, w) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1) + res)
    fd = P.fwd_desc(geom, N, H, W)
    packer = WeightPacker()
    packer.add(0, 128, 1, 1, 64, fd.pack)
    packer.build(DEV, dtype).run(w.permute(0, 2, 3, 1).contiguous().view(-1).to(DEV))
    xa = nhwc(x).to(DEV).to(dtype)
    y = torch.empty(N, H, W, 128, dtype=dtype, device=DEV)
    ops.conv_igemm(fd, xa, packer.view(fd.pack, 128), y, scale=scale.to(DEV), shift=shift.to(DEV),
                   residual=nhwc(res).to(DEV).to(dtype), relu=True)
    assert relmax(y.float(), nhwc(ref)) < tol(dtype)
    # fp32 output, shift only (Linear + bias)
    y32 = torch.empty(N, H, W, 128, dtype=torch.float32, device=DEV)
    ops.conv_igemm(fd, xa, packer.view(fd.pack, 128), y32, shift=shift.to(DEV), out_f32=True)
    ref2 = F.conv2d(x, w) + shift.view(1, -1, 1, 1)
    assert relmax(y32, nhwc(ref2)) < (1e-4 if dtype == torch.float32 else 1e-3)


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('hw', [(64, 64), (33, 47)])
def test_stem_conv_and_wgrad(dtype, hw):
    H, W = hw
    N, cout = 4, 64
    gen = torch.Generator().manual_seed(5)
    x = rnd(torch.randn(N, 3, H, W, generator=gen), dtype).requires_grad_(True)
    w = rnd(torch.randn(cout, 3, 7, 7, generator=gen) * 0.1, dtype).requires_grad_(True)
    y = F.conv2d(x.double(), w.double(), None, 2, 3)
    dy = rnd(torch.randn(y.shape, generator=gen), dtype)
    y.backward(dy.double())
    d = P.stem_desc(cout, N, H, W)
    Hp, Wp = P.stem_padded_hw(H, W)
    xp = ops.nchw_to_nhwc_pad(x.detach().to(DEV), P.STEM_PAD, Wp, P.STEM_CP, dtype)
    packer = WeightPacker()
    packer.add(0, cout, 7, 7, 3, d.pack)
    packer.build(DEV, dtype).run(w.detach().permute(0, 2, 3, 1).contiguous().view(-1).to(DEV))
    yo = torch.empty(N, d.OP, d.OQ, cout, dtype=dtype, device=DEV)
    ops.conv_igemm(d, xp, packer.view(d.pack, cout), yo)
    assert relmax(yo.float(), nhwc(y.detach())) < tol(dtype)
    dwp = torch.zeros(cout, 7 * 32, dtype=torch.float32, device=DEV)
    ops.conv_wgrad(d, xp, nhwc(dy).to(DEV).to(dtype).view(-1, cout), dwp)
    got = dwp.view(cout, 7, 8, 4)[:, :, :7, :3]
    assert relmax(got, w.grad.permute(0, 2, 3, 1)) < (1e-4 if dtype == torch.float32 else 2e-3)


@pytest.mark.parametrize('hw', [(224, 224), (64, 96)])
def test_stem_kernel_against_generic_kernel(hw):
    """The spatially tiled stem kernel (csrc/conv_stem.hip) walks K in the same 32-wide steps as the
    generic implicit-GEMM kernel: outputs must agree BIT FOR BIT (plain, and with the fused affine + ReLU
    of the key encoder); its fused BatchNorm statistics (one slab row per 8 x 16 tile) must finalize to the
    statistics of a separate pass over the stored output."""
    from passl_amd.hip import lib as L
    lib = L.load()
    dtype = torch.bfloat16
    H, W = hw
    N, cout = 3, 64
    gen = torch.Generator().manual_seed(17)
    x = rnd(torch.randn(N, 3, H, W, generator=gen) * 1.3 + 0.2, dtype)
    w = rnd(torch.randn(cout, 3, 7, 7, generator=gen) * 0.1, dtype)
    d = P.stem_desc(cout, N, H, W)
    _Hp, Wp = P.stem_padded_hw(H, W)
    xp = ops.nchw_to_nhwc_pad(x.to(DEV), P.STEM_PAD, Wp, P.STEM_CP, dtype)
    packer = WeightPacker()
    packer.add(0, cout, 7, 7, 3, d.pack)
    packer.build(DEV, dtype).run(w.permute(0, 2, 3, 1).contiguous().view(-1).to(DEV))
    scale = (torch.rand(cout, generator=gen) + 0.5).to(DEV)
    shift = torch.randn(cout, generator=gen).to(DEV)

    def run():
        y = torch.full((N, d.OP, d.OQ, cout), float('nan'), dtype=dtype, device=DEV)
        slab, tiles = ops.conv_stats_buffer(d, DEV)
        slab.fill_(float('nan'))
        ops.conv_igemm(d, xp, packer.view(d.pack, cout), y, stats=slab)
        y2 = torch.full_like(y, float('nan'))
        ops.conv_igemm(d, xp, packer.view(d.pack, cout), y2, scale=scale, shift=shift, relu=True)
        g, b = torch.ones(cout, device=DEV), torch.zeros(cout, device=DEV)
        rm, rv = torch.zeros(cout, device=DEV), torch.ones(cout, device=DEV)
        _z, st, _ = ops.bn_train_fwd(y, g, b, rm, rv, relu=False, partial=(slab, tiles))
        return y, y2, st, rm, rv
    try:
        assert lib.passl_hip_set_option(b'stem_kernel', 0) == 0
        y0, y20, st0, rm0, rv0 = run()
        assert lib.passl_hip_set_option(b'stem_kernel', 1) == 0
        y1, y21, st1, rm1, rv1 = run()
    finally:
        lib.passl_hip_set_option(b'stem_kernel', 1)
    ref = F.conv2d(x.double(), w.double(), None, 2, 3)
    assert relmax(y1.float(), nhwc(ref)) < tol(dtype)
    assert torch.equal(y0, y1) and torch.equal(y20, y21)
    assert relmax(st1[0], st0[0]) < 1e-5 and relmax(st1[1], st0[1]) < 1e-5
    assert relmax(rm1, rm0) < 1e-5 and relmax(rv1, rv0) < 1e-5
    yf = y1.float().view(-1, cout).double()
    assert relmax(st1[0], yf.mean(0)) < 1e-5
    assert relmax(st1[1], (yf.var(0, unbiased=False) + 1e-5).rsqrt()) < 1e-5


@pytest.mark.parametrize('dtype', DTYPES)
def test_linear_as_conv(dtype):
    """Linear 2048->128 with bias on [256, 2048] rows (projector), fwd + dgrad + wgrad."""
    N, cin, cout = 256, 2048, 128
    gen = torch.Generator().manual_seed(9)
    x = rnd(torch.randn(N, cin, generator=gen), dtype)
    w = rnd(torch.randn(cout, cin, generator=gen) * 0.02, dtype)
    b = torch.randn(cout, generator=gen)
    geom = P.ConvGeom(cin, cout, 1, 1, 0)
    fd = P.fwd_desc(geom, N, 1, 1)
    dds, _ = P.dgrad_plan(geom, N, 1, 1)
    packer = WeightPacker()
    for d in [fd] + dds:
        packer.add(0, cout, 1, 1, cin, d.pack)
    packer.build(DEV, dtype).run(w.contiguous().view(-1).to(DEV))
    xa = x.to(DEV).to(dtype)
    y = torch.empty(N, cout, dtype=torch.float32, device=DEV)
    ops.conv_igemm(fd, xa, packer.view(fd.pack, cout), y, shift=b.to(DEV), out_f32=True)
    assert relmax(y, x.double() @ w.double().t() + b.double()) < (1e-4 if dtype == torch.float32 else 2e-3)
    dy = rnd(torch.randn(N, cout, generator=gen), dtype)
    dx = torch.empty(N, cin, dtype=dtype, device=DEV)
    ops.conv_igemm(dds[0], dy.to(DEV).to(dtype), packer.view(dds[0].pack, cin), dx)
    assert relmax(dx.float(), dy.double() @ w.double()) < tol(dtype)
    dw = torch.zeros(cout, cin, dtype=torch.float32, device=DEV)
    ops.conv_wgrad(P.wgrad_desc(geom, N, 1, 1), xa, dy.to(DEV).to(dtype), dw)
    assert relmax(dw, dy.double().t() @ x.double()) < (1e-4 if dtype == torch.float32 else 2e-3)
    db = torch.empty(cout, dtype=torch.float32, device=DEV)
    ops.colsum_into(dy.to(DEV).to(dtype), db)
    assert relmax(db, dy.double().sum(0)) < 1e-5


def test_bn_fold():
    """Inference-form BatchNorm affine of all layers in one launch against the torch formula."""
    gen = torch.Generator().manual_seed(3)
    flat = torch.randn(5000, generator=gen)
    flat[3000:4000] = flat[3000:4000].abs() + 0.1            # variances
    gi, bi = torch.arange(0, 1000), torch.arange(1000, 2000)
    mi, vi = torch.arange(2000, 3000), torch.arange(3000, 4000)
    idx = tuple(t.to(DEV) for t in (gi, bi, mi, vi))
    scale, shift = torch.empty(1000, device=DEV), torch.empty(1000, device=DEV)
    ops.bn_fold(flat.to(DEV), idx, 1e-5, scale, shift)
    ref_s = flat[gi].double() / (flat[vi].double() + 1e-5).sqrt()
    ref_b = flat[bi].double() - flat[mi].double() * ref_s
    assert relmax(scale, ref_s) < 1e-6 and relmax(shift, ref_b) < 1e-6


# ---------------------------------------------------------------- batch norm
@pytest.mark.parametrize('dtype', DTYPES)
# (5, 125, 128, 256): 625 row slabs -> the finalize kernels combine the slab in segments
@pytest.mark.parametrize('shape', [(4, 56, 56, 64), (2, 7, 7, 2048), (3, 5, 5, 24), (8, 14, 14, 256),
                                   (5, 125, 128, 256)])
@pytest.mark.parametrize('with_res', [False, True])
def test_bn_fwd_bwd(dtype, shape, with_res):
    N, H, W, Cc = shape
    gen = torch.Generator().manual_seed(11)
    x = rnd(torch.randn(N, H, W, Cc, generator=gen) * 2 + 0.5, dtype).requires_grad_(True)
    gamma = (torch.rand(Cc, generator=gen) + 0.5).requires_grad_(True)
    beta = torch.randn(Cc, generator=gen).requires_grad_(True)
    res = rnd(torch.randn(N, H, W, Cc, generator=gen), dtype).requires_grad_(True)
    rm0 = torch.randn(Cc, generator=gen); rv0 = torch.rand(Cc, generator=gen) + 0.5
    x64 = x.double()
    mu = x64.mean(dim=(0, 1, 2)); var = x64.var(dim=(0, 1, 2), unbiased=False)
    xh = (x64 - mu) / torch.sqrt(var + 1e-5)
    pre = xh * gamma.double() + beta.double()
    if with_res:
        pre = pre + res.double()
    z_ref = F.relu(pre)
    dz = rnd(torch.randn(shape, generator=gen), dtype)
    z_ref.backward(dz.double())
    rm, rv = rm0.clone().to(DEV), rv0.clone().to(DEV)
    xd = x.detach().to(DEV).to(dtype)
    resd = res.detach().to(DEV).to(dtype) if with_res else None
    z, st, mask = ops.bn_train_fwd(xd, gamma.detach().to(DEV), beta.detach().to(DEV), rm, rv,
                                   residual=resd, relu=True, want_mask=True)
    mean, invstd = st[0], st[1]
    assert relmax(z.float(), z_ref.detach()) < tol(dtype)
    # the bit mask is exactly (z > 0), bit e of byte i <-> element 8*i + e
    bits = (z.reshape(-1, 8) > 0).to(torch.int32) * (1 << torch.arange(8, device=DEV, dtype=torch.int32))
    assert torch.equal(bits.sum(1).to(torch.uint8), mask)
    assert relmax(mean, mu) < 1e-5 and relmax(invstd, 1 / torch.sqrt(var + 1e-5)) < 1e-5
    assert relmax(rm, 0.9 * rm0 + 0.1 * mu) < 1e-5
    assert relmax(rv, 0.9 * rv0 + 0.1 * var) < 1e-5
    dgamma = torch.zeros(Cc, device=DEV); dbeta = torch.zeros(Cc, device=DEV)
    # the mask must come from the product's own z (bf16 rounding can flip tiny values)
    dx, dres = ops.bn_bwd(dz.to(DEV).to(dtype), z, xd, gamma.detach().to(DEV), mean, invstd,
                          dgamma, dbeta, relu=True, want_dres=with_res)
    t = 1e-4 if dtype == torch.float32 else 3e-2
    # elements whose pre-activation is within rounding of zero may take either side of the ReLU (the
    # product evaluates x*scale + shift + res in fp32, the reference in float64): with 20 M elements a
    # handful always do, and there dx legitimately differs by a whole gradient value — leave them out
    sure = (pre.detach().abs() > 1e-5 * float(pre.detach().abs().max())).to(DEV)
    xg = x.grad.to(DEV)
    assert float(sure.double().mean()) > 0.999
    assert relmax(torch.where(sure, dx.double(), xg), xg) < t
    # (one ReLU-ambiguous element moves a channel's sum by a whole gradient value: visible at 1e-4 only
    # in the tall case)
    tsum = 1e-2 if (dtype != torch.float32 or not bool(sure.all())) else 1e-4
    assert relmax(dgamma, gamma.grad) < tsum
    assert relmax(dbeta, beta.grad) < tsum
    if with_res:
        rg = res.grad.to(DEV)
        assert relmax(torch.where(sure, dres.double(), rg), rg) < t
    # the other mask sources must reproduce the z-based result bit for bit: 3 = bit mask,
    # 2 = recomputed from x*scale+shift (BN+ReLU without residual only)
    for mode in ([3] if with_res else [3, 2]):
        dg2 = torch.zeros(Cc, device=DEV); db2 = torch.zeros(Cc, device=DEV)
        dx2, dres2 = ops.bn_bwd(dz.to(DEV).to(dtype), mask if mode == 3 else None, xd,
                                gamma.detach().to(DEV), mean, invstd, dg2, db2, relu=mode,
                                want_dres=with_res, scale=st[2], shift=st[3])
        assert torch.equal(dx2, dx) and torch.equal(dg2, dgamma) and torch.equal(db2, dbeta)
        if with_res:
            assert torch.equal(dres2, dres)


@pytest.mark.parametrize('mode', [2, 3])
@pytest.mark.parametrize('geom,nhw', [
    (P.ConvGeom(64, 256, 1, 1, 0), (2, 28, 28)),      # conv3-style 1x1: dgrad is a dense 1x1
    (P.ConvGeom(64, 64, 3, 1, 1), (3, 14, 14)),       # conv2 stride 1: one flipped-filter launch
    (P.ConvGeom(128, 128, 3, 2, 1), (3, 28, 28)),     # conv2 stride 2: four residue-class launches
    (P.ConvGeom(512, 128, 1, 1, 0), (2, 7, 7)),       # ring kernel, ragged M tile (M = 98)
])
def test_dgrad_fused_bn_backward(geom, nhw, mode):
    """BatchNorm-backward statistics fused into the data-gradient launch (csrc/igemm_epi.h) against the
    separate path: the stored gradient must be the masked gradient BIT FOR BIT, the slab must give
    the same (dgamma, dbeta, dx) as bn_bwd_reduce to summation-order accuracy, reproducibly."""
    dtype = torch.bfloat16
    N, H, W = nhw
    gen = torch.Generator().manual_seed(29)
    Cin = geom.cin
    # the BatchNorm in front of the conv: y -> z = relu(bn(y) [+ res]) -> conv
    y = rnd(torch.randn(N, H, W, Cin, generator=gen) * 1.5 + 0.3, dtype).to(DEV).to(dtype)
    res = rnd(torch.randn(N, H, W, Cin, generator=gen), dtype).to(DEV).to(dtype) if mode == 3 else None
    gamma = (torch.rand(Cin, generator=gen) + 0.5).to(DEV)
    beta = torch.randn(Cin, generator=gen).to(DEV)
    rm, rv = torch.zeros(Cin, device=DEV), torch.ones(Cin, device=DEV)
    z, st, mask = ops.bn_train_fwd(y, gamma, beta, rm, rv, residual=res, relu=True, want_mask=(mode == 3))
    w = rnd(torch.randn(geom.cout, geom.cin, geom.k, geom.k, generator=gen) * 0.1, dtype)
    dds, skipped = P.dgrad_plan(geom, N, H, W)
    assert not skipped
    packer = WeightPacker()
    for d in dds:
        packer.add(0, geom.cout, geom.k, geom.k, geom.cin, d.pack)
    packer.build(DEV, dtype).run(w.permute(0, 2, 3, 1).contiguous().to(DEV).view(-1))
    OP, OQ = geom.out_hw(H, W)
    dy = rnd(torch.randn(N, OP, OQ, geom.cout, generator=gen), dtype).to(DEV).to(dtype)
    extra = None
    if len(dds) == 1:
        extra = rnd(torch.randn(N, H, W, Cin, generator=gen), dtype).to(DEV).to(dtype)
    # separate path: plain dgrad, then the BatchNorm backward masks and reduces
    dz = torch.empty(N, H, W, Cin, dtype=dtype, device=DEV)
    for d in dds:
        ops.conv_igemm(d, dy, packer.view(d.pack, Cin), dz, residual=extra)
    dg0, db0 = torch.zeros(Cin, device=DEV), torch.zeros(Cin, device=DEV)
    dx0, dres0 = ops.bn_bwd(dz, mask, y, gamma, st[0], st[1], dg0, db0, relu=mode, want_dres=(mode == 3),
                            scale=st[2], shift=st[3])
    # fused path
    def fused_run():
        tiles = sum(ops.conv_tiles(d) for d in dds)
        slab = torch.full((ops.bn_partial_floats(tiles, Cin, False),), float('nan'), dtype=torch.float32,
                          device=DEV)
        g = torch.full((N, H, W, Cin), float('nan'), dtype=dtype, device=DEV)
        off = 0
        for d in dds:
            ops.conv_igemm(d, dy, packer.view(d.pack, Cin), g, residual=extra,
                           bnb=dict(y=y, mask=mask, mean=st[0], invstd=st[1], scale=st[2], shift=st[3],
                                    relu=mode, partial=slab, tile_off=off))
            off += ops.conv_tiles(d)
        return g, slab, tiles
    g, slab, tiles = fused_run()
    assert not torch.isnan(slab[:tiles * Cin * 2]).any() and not torch.isnan(g.float()).any()
    keep = (z > 0)
    assert torch.equal(g, torch.where(keep, dz, torch.zeros_like(dz)))
    dg1, db1 = torch.zeros(Cin, device=DEV), torch.zeros(Cin, device=DEV)
    dx1, dres1 = ops.bn_bwd(g, None, y, gamma, st[0], st[1], dg1, db1, relu=mode, want_dres=(mode == 3),
                            scale=st[2], shift=st[3], fused=(slab, tiles))
    assert relmax(dg1, dg0) < 2e-5 and relmax(db1, db0) < 2e-5
    assert relmax(dx1.float(), dx0.float()) < 1e-2          # bf16 outputs of coefficients that differ by 1e-5
    if mode == 3:
        assert dres1 is g and torch.equal(dres1, dres0)      # the residual-branch gradient IS g: no copy
    g2, slab2, _ = fused_run()
    assert torch.equal(g2, g) and torch.equal(slab2[:tiles * Cin * 2], slab[:tiles * Cin * 2])   # reproducible


@pytest.mark.parametrize('geom,nhw', [
    (P.ConvGeom(256, 64, 1, 1, 0), (4, 28, 28)),      # stage-1 shape: data gradient K = 64 -> 256 columns, one K-tile
    (P.ConvGeom(512, 128, 1, 1, 0), (3, 14, 14)),     # two K-tiles
    (P.ConvGeom(2048, 512, 1, 1, 0), (2, 7, 7)),      # eight K-tiles: the ring kernel's instantiation, ragged M tile
])
def test_dgrad_fused_two_batchnorms(geom, nhw):
    """A bottleneck block with a downsample branch ends in relu(bn3(y3) + bn_ds(y_ds)) (resnetimagenet.py:139-153): the
    masked output gradient g is the output gradient of BOTH BatchNorm layers.  The data-gradient launch that produces g
    (conv1 of the next block) writes the (sum g, sum g * xhat) slab of bn3 and — passl_conv_desc.bnb2_* — of bn_ds in
    one pass; against the separate passes: g bit for bit, both layers' (d-gamma, d-beta, dx) to summation-order
    accuracy, reproducibly."""
    dtype = torch.bfloat16
    N, H, W = nhw
    gen = torch.Generator().manual_seed(31)
    C = geom.cin

    def t(*shape, scale=1.0, shift=0.0):
        return rnd(torch.randn(*shape, generator=gen) * scale + shift, dtype).to(DEV).to(dtype)
    y3, yds = t(N, H, W, C, scale=1.5, shift=0.3), t(N, H, W, C, scale=0.7, shift=-0.2)
    g3, b3 = (torch.rand(C, generator=gen) + 0.5).to(DEV), torch.randn(C, generator=gen).to(DEV)
    gd, bd = (torch.rand(C, generator=gen) + 0.5).to(DEV), torch.randn(C, generator=gen).to(DEV)
    rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    idn, st_ds, _ = ops.bn_train_fwd(yds, gd, bd, rm.clone(), rv.clone(), relu=False)
    z, st3, mask = ops.bn_train_fwd(y3, g3, b3, rm.clone(), rv.clone(), residual=idn, relu=True, want_mask=True)
    w = rnd(torch.randn(geom.cout, geom.cin, 1, 1, generator=gen) * 0.1, dtype)
    dds, skipped = P.dgrad_plan(geom, N, H, W)
    assert not skipped and len(dds) == 1
    d = dds[0]
    packer = WeightPacker()
    packer.add(0, geom.cout, 1, 1, geom.cin, d.pack)
    packer.build(DEV, dtype).run(w.permute(0, 2, 3, 1).contiguous().to(DEV).view(-1))
    dy = t(N, H, W, geom.cout)
    extra = t(N, H, W, C)                          # the gradient of the block's identity path (GradSlot)
    # separate passes
    dz = torch.empty(N, H, W, C, dtype=dtype, device=DEV)
    ops.conv_igemm(d, dy, packer.view(d.pack, C), dz, residual=extra)
    dg3, db3 = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    dx3, dres = ops.bn_bwd(dz, mask, y3, g3, st3[0], st3[1], dg3, db3, relu=3, want_dres=True, scale=st3[2], shift=st3[3])
    dgd, dbd = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    dxd, _ = ops.bn_bwd(dres, None, yds, gd, st_ds[0], st_ds[1], dgd, dbd, relu=0)

    def fused_run():
        tiles = ops.conv_tiles(d)
        nan = float('nan')
        slab = torch.full((ops.bn_partial_floats(tiles, C, False),), nan, dtype=torch.float32, device=DEV)
        slab2 = torch.full((ops.bn_partial_floats(tiles, C, False),), nan, dtype=torch.float32, device=DEV)
        g = torch.full((N, H, W, C), nan, dtype=dtype, device=DEV)
        ops.conv_igemm(d, dy, packer.view(d.pack, C), g, residual=extra,
                       bnb=dict(y=y3, mask=mask, mean=st3[0], invstd=st3[1], scale=st3[2], shift=st3[3], relu=3,
                                partial=slab, tile_off=0, y2=yds, mean2=st_ds[0], invstd2=st_ds[1], partial2=slab2))
        return g, slab, slab2, tiles
    g, slab, slab2, tiles = fused_run()
    n = tiles * C * 2
    assert not torch.isnan(slab[:n]).any() and not torch.isnan(slab2[:n]).any() and not torch.isnan(g.float()).any()
    assert torch.equal(g, dres)                                       # the masked gradient, bit for bit
    assert torch.equal(slab[:n].view(tiles, C, 2)[..., 0], slab2[:n].view(tiles, C, 2)[..., 0])    # sum g is shared
    f3g, f3b = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    fx3, _ = ops.bn_bwd(g, None, y3, g3, st3[0], st3[1], f3g, f3b, relu=3, want_dres=True, scale=st3[2], shift=st3[3],
                        fused=(slab, tiles))
    fdg, fdb = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    fxd, _ = ops.bn_bwd(g, None, yds, gd, st_ds[0], st_ds[1], fdg, fdb, relu=0, fused=(slab2, tiles))
    for a, b in ((f3g, dg3), (f3b, db3), (fdg, dgd), (fdb, dbd)):
        assert relmax(a, b) < 2e-5
    assert relmax(fx3.float(), dx3.float()) < 1e-2 and relmax(fxd.float(), dxd.float()) < 1e-2
    g_b, slab_b, slab2_b, _ = fused_run()
    assert torch.equal(g_b, g) and torch.equal(slab_b[:n], slab[:n]) and torch.equal(slab2_b[:n], slab2[:n])


@pytest.mark.parametrize('dtype', DTYPES)
def test_bn_statistics_large_mean(dtype):
    """|mean| >> std: E[x^2] - mean^2 from fp32 sums would lose the variance; the shifted sums do not."""
    gen = torch.Generator().manual_seed(5)
    M, Cc = 4096, 16
    x = rnd(torch.randn(M, Cc, generator=gen) * 0.25 + 200.0, dtype)
    xd = x.to(DEV).to(dtype)
    ga, be = torch.ones(Cc, device=DEV), torch.zeros(Cc, device=DEV)
    rm, rv = torch.zeros(Cc, device=DEV), torch.ones(Cc, device=DEV)
    _z, st, _m = ops.bn_train_fwd(xd, ga, be, rm, rv, relu=False)
    x64 = x.double()
    assert relmax(st[0], x64.mean(0)) < 1e-6
    assert relmax(st[1], 1 / torch.sqrt(x64.var(0, unbiased=False) + 1e-5)) < 1e-5


# ---------------------------------------------------------------- pooling
@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('shape', [(2, 64, 112, 112), (3, 64, 33, 17), (1, 8, 5, 5)])
def test_maxpool(dtype, shape):
    gen = torch.Generator().manual_seed(13)
    x = rnd(torch.randn(shape, generator=gen), dtype).requires_grad_(True)
    y = F.max_pool2d(x, 3, 2, 1)
    dy = rnd(torch.randn(y.shape, generator=gen), dtype)
    y.backward(dy)
    yo, idx = ops.maxpool_fwd(nhwc(x.detach()).to(DEV).to(dtype))
    assert (yo.float().cpu() - nhwc(y.detach())).abs().max() == 0
    dx = ops.maxpool_bwd(nhwc(dy).to(DEV).to(dtype), idx, shape[2], shape[3])
    assert relmax(dx.float(), nhwc(x.grad)) < (1e-6 if dtype == torch.float32 else 1e-2)


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('shape', [(4, 32, 32, 64), (3, 17, 21, 64), (2, 9, 8, 256), (256, 112, 112, 64)])
def test_bn_relu_maxpool_fused_equals_the_separate_passes(dtype, shape):
    """csrc/stem_pool.hip (BatchNorm + ReLU + MaxPool2D(3, 2, 1) of the stem, resnetimagenet.py:196-198, one pass per
    direction) against the passes it replaces, fed with the same tensors: the pooled output and the index bytes are
    bit-identical, the input gradient is bit-identical GIVEN THE SAME coefficients (checked by handing the fused apply
    pass the separate path's coefficients), and the fused reduce pass's per-channel sums agree to fp32 rounding."""
    if shape[0] == 256 and dtype == torch.float32:
        pytest.skip('full stem size in the benchmark dtype only')
    from passl_amd.hip import lib as L
    gen = torch.Generator().manual_seed(131)
    N, H, W, C = shape
    x = rnd(torch.randn(shape, generator=gen) * 1.5 + 0.3, dtype).to(DEV).to(dtype)
    gamma = (torch.rand(C, generator=gen) + 0.5).to(DEV)
    beta = (torch.randn(C, generator=gen) * 0.2).to(DEV)
    rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    # the separate passes
    z, st, _ = ops.bn_train_fwd(x, gamma, beta, rm.clone(), rv.clone(), None, True)
    y_ref, idx_ref = ops.maxpool_fwd(z)
    dy = rnd(torch.randn(y_ref.shape, generator=gen), dtype).to(DEV).to(dtype)
    dz = ops.maxpool_bwd(dy, idx_ref, H, W)
    dg_ref, db_ref = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    dx_ref, _ = ops.bn_bwd(dz, None, x, gamma, st[0], st[1], dg_ref, db_ref, relu=2, scale=st[2], shift=st[3])
    # fused
    assert ops.bn_relu_maxpool_supported(x)
    _, st2, _ = ops.bn_train_fwd(x, gamma, beta, rm.clone(), rv.clone(), None, True, apply=False)
    assert torch.equal(st, st2)
    y, idx = ops.bn_relu_maxpool_fwd(x, st2)
    assert torch.equal(y, y_ref) and torch.equal(idx, idx_ref)
    dg, db = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    dx = ops.bn_relu_maxpool_bwd(dy, idx, x, gamma, st2, dg, db)
    tol = 2e-6 if dtype == torch.float32 else 1e-5       # sums of ~N*H*W terms added in another order
    assert relmax(dg, dg_ref) < tol and relmax(db, db_ref) < tol
    assert relmax(dx.float(), dx_ref.float()) < (1e-5 if dtype == torch.float32 else 1e-2)
    # the apply pass with the separate path's coefficients: every element the same bits
    coef = torch.empty(3 * C, dtype=torch.float32, device=DEV)
    lib = L.load()
    nb = lib.passl_hip_bn_relu_maxpool_blocks(N, H, W, C)
    part = torch.empty(ops.bn_partial_floats(nb, C, False), dtype=torch.float32, device=DEV)
    L.check(lib.passl_hip_bn_bwd_reduce(L.ptr(dz), None, L.ptr(x), L.ptr(st[0]), L.ptr(st[1]), L.ptr(st[2]), L.ptr(st[3]),
                                        L.ptr(part), N * H * W, C, nb, 2, L.dt(x), L.stream()), 'bn_bwd_reduce')
    scratch = [torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)]
    L.check(lib.passl_hip_bn_bwd_finalize(L.ptr(part), nb, N * H * W, C, L.ptr(gamma), L.ptr(st[0]), L.ptr(st[1]),
                                          L.ptr(scratch[0]), L.ptr(scratch[1]), L.ptr(coef), L.stream()), 'bn_bwd_finalize')
    dx_a = torch.empty_like(x)
    L.check(lib.passl_hip_bn_bwd_apply(L.ptr(dz), None, L.ptr(x), L.ptr(coef), L.ptr(st[2]), L.ptr(st[3]), L.ptr(dx_a), None,
                                       N * H * W, C, 2, L.dt(x), L.stream()), 'bn_bwd_apply')
    dx_b = torch.empty_like(x)
    L.check(lib.passl_hip_bn_relu_maxpool_bwd_apply(L.ptr(dy), L.ptr(idx), L.ptr(x), L.ptr(coef), L.ptr(st[2]), L.ptr(st[3]),
                                                    L.ptr(dx_b), N, H, W, C, L.dt(x), L.stream()), 'fused apply')
    assert torch.equal(dx_a, dx_b)


@pytest.mark.parametrize('dtype', DTYPES)
def test_avgpool(dtype):
    gen = torch.Generator().manual_seed(14)
    x = rnd(torch.randn(5, 7, 7, 2048, generator=gen), dtype)
    y = ops.avgpool_fwd(x.to(DEV).to(dtype))
    assert relmax(y.float(), x.mean(dim=(1, 2))) < (1e-6 if dtype == torch.float32 else 1e-2)
    dy = rnd(torch.randn(5, 2048, generator=gen), dtype)
    dx = ops.avgpool_bwd(dy.to(DEV).to(dtype), 7, 7)
    ref = (dy / 49.0).view(5, 1, 1, 2048).expand(5, 7, 7, 2048)
    assert relmax(dx.float(), ref) < (1e-6 if dtype == torch.float32 else 1e-2)


# ---------------------------------------------------------------- head
def test_l2norm_fwd_bwd():
    gen = torch.Generator().manual_seed(15)
    x = torch.randn(37, 128, generator=gen, requires_grad=True)
    y = x / x.norm(dim=1, keepdim=True).clamp_min(1e-12)
    dy = torch.randn(37, 128, generator=gen)
    y.backward(dy)
    yo, norm = ops.l2norm_fwd(x.detach().to(DEV))
    assert relmax(yo, y.detach()) < 1e-6
    dx = ops.l2norm_bwd(dy.to(DEV), yo, norm, torch.float32)
    assert relmax(dx, x.grad) < 1e-5
    dxb = ops.l2norm_bwd(dy.to(DEV), yo, norm, torch.bfloat16)
    assert relmax(dxb.float(), x.grad) < 1e-2


@pytest.mark.parametrize('N,K', [(256, 65536), (32, 65536), (8, 1024), (70, 2048)])
def test_infonce_fwd_bwd(N, K):
    gen = torch.Generator().manual_seed(16)
    T = 0.2
    q = F.normalize(torch.randn(N, 128, generator=gen), dim=1).requires_grad_(True)
    k = F.normalize(q.detach() + 0.5 * torch.randn(N, 128, generator=gen), dim=1)
    queue = F.normalize(torch.randn(128, K, generator=gen), dim=0)
    # make a few negatives beat the positive so acc1/acc5 are non-trivial
    queue[:, 3] = q.detach()[0]; queue[:, 9] = q.detach()[1]
    q64 = q.double()
    logits = torch.cat([(q64 * k.double()).sum(1, keepdim=True), q64 @ queue.double()], 1) / T
    loss = F.cross_entropy(logits, torch.zeros(N, dtype=torch.long))
    loss.backward()
    rank = (logits[:, 1:] > logits[:, :1]).sum(1)
    out, lse, lg = ops.infonce_fwd(q.detach().to(DEV), k.to(DEV), queue.to(DEV), T, want_logits=True)
    out = out.cpu()
    assert abs(out[0].item() - loss.item()) < 2e-5 * max(1.0, abs(loss.item()))
    assert abs(out[1].item() - (rank < 1).double().mean().item() * 100) < 1e-3
    assert abs(out[2].item() - (rank < 5).double().mean().item() * 100) < 1e-3
    assert (lg.double().cpu() - logits.detach()).abs().max() < 1e-4
    assert (lse.double().cpu() - torch.logsumexp(logits.detach(), 1)).abs().max() < 1e-4
    gs = torch.tensor([1.0], device=DEV)
    dq = ops.infonce_bwd(q.detach().to(DEV), k.to(DEV), queue.to(DEV), lse, gs, T)
    assert relmax(dq, q.grad) < 1e-4
    dq2 = ops.infonce_bwd(q.detach().to(DEV), k.to(DEV), queue.to(DEV), lse, None, T)
    assert relmax(dq2, q.grad) < 1e-4
    # no atomics anywhere (per-slice slabs + fixed-order sums): bit-reproducible loss and gradient
    out_b, lse_b, _ = ops.infonce_fwd(q.detach().to(DEV), k.to(DEV), queue.to(DEV), T)
    assert torch.equal(out_b.cpu(), out) and torch.equal(lse_b, lse)
    assert torch.equal(ops.infonce_bwd(q.detach().to(DEV), k.to(DEV), queue.to(DEV), lse, gs, T), dq)


def test_enqueue():
    gen = torch.Generator().manual_seed(17)
    queue = torch.randn(128, 1024, generator=gen)
    keys = torch.randn(32, 128, generator=gen)
    qd = queue.clone().to(DEV)
    ops.enqueue(qd, keys.to(DEV), 96)
    queue[:, 96:128] = keys.t()
    assert (qd.cpu() - queue).abs().max() == 0


@pytest.mark.parametrize('k', [1, 3])
def test_conv_operands_beyond_2gb(k):
    """Tensors larger than a buffer descriptor's 32-bit byte window (BASELINE configs[2] at bs 512 per GPU:
    1024 x 112^2 x 256 channels = 6.6 GB): the LDS-DMA kernels rebase their descriptors per workgroup /
    per M-slice.  The same launch over the whole batch and over two halves (each below 2 GB, i.e. the
    classic addressing) must agree — bit for bit for the forward (same tiles, same order)."""
    dtype = torch.bfloat16
    N, H, Cin, Cout = 176, 112, 512, 64                    # x: 176*112*112*512*2 B = 2.26 GB
    geom = P.ConvGeom(Cin, Cout, k, 1, k // 2)
    gen = torch.Generator().manual_seed(41)
    x = torch.randn(N, H, H, Cin, device=DEV, dtype=dtype)
    assert x.numel() * 2 > (1 << 31)
    w = (torch.randn(Cout, k, k, Cin, generator=gen) * 0.05).to(DEV)
    fd = P.fwd_desc(geom, N, H, H)
    fh = P.fwd_desc(geom, N // 2, H, H)
    packer = WeightPacker()
    packer.add(0, Cout, k, k, Cin, fd.pack)
    packer.build(DEV, dtype).run(w.view(-1))
    wb = packer.view(fd.pack, Cout)
    y = torch.empty(N, H, H, Cout, device=DEV, dtype=dtype)
    ops.conv_igemm(fd, x, wb, y)
    yh = torch.empty_like(y)
    ops.conv_igemm(fh, x[:N // 2], wb, yh[:N // 2])
    ops.conv_igemm(fh, x[N // 2:], wb, yh[N // 2:])
    assert torch.equal(y, yh)
    # weight gradient: dy small, x > 2 GB
    dy = torch.randn(N, H, H, Cout, device=DEV, dtype=dtype)
    dw = torch.zeros(Cout, k * k * Cin, device=DEV)
    ops.conv_wgrad(P.wgrad_desc(geom, N, H, H), x, dy.view(-1, Cout), dw)
    dwh = torch.zeros_like(dw)
    wdh = P.wgrad_desc(geom, N // 2, H, H)
    ops.conv_wgrad(wdh, x[:N // 2], dy[:N // 2].reshape(-1, Cout), dwh)
    ops.conv_wgrad(wdh, x[N // 2:], dy[N // 2:].reshape(-1, Cout), dwh)
    assert relmax(dw, dwh) < 1e-3
    # data gradient of the transposed shape: the OUTPUT (dx) is the > 2 GB tensor, rows addressed 64-bit
    if k == 1:
        g2 = P.ConvGeom(Cout, Cin, 1, 1, 0)                  # conv 64 -> 512: dgrad reads dy2 [.,512] > 2 GB
        dds, _ = P.dgrad_plan(g2, N, H, H)
        dh, _ = P.dgrad_plan(g2, N // 2, H, H)
        pk = WeightPacker()
        for d in dds + dh:
            pk.add(0, Cin, 1, 1, Cout, d.pack)
        w2 = (torch.randn(Cin, 1, 1, Cout, generator=gen) * 0.05).to(DEV)
        pk.build(DEV, dtype).run(w2.view(-1))
        dx = torch.empty(N, H, H, Cout, device=DEV, dtype=dtype)
        dxh = torch.empty_like(dx)
        ops.conv_igemm(dds[0], x, pk.view(dds[0].pack, Cout), dx)            # x plays dy2 (2.05 GB, K = 512: ring)
        ops.conv_igemm(dh[0], x[:N // 2], pk.view(dh[0].pack, Cout), dxh[:N // 2])
        ops.conv_igemm(dh[0], x[N // 2:], pk.view(dh[0].pack, Cout), dxh[N // 2:])
        assert torch.equal(dx, dxh)


def test_fill_copy_cast_unpad_kernels():
    """The library's own fills / copies / casts (csrc/plan.hip): what replaces the step's last ATen launches so that
    a recorded step plan holds library launches only — all sizes incl. unaligned heads / tails."""
    gen = torch.Generator().manual_seed(5)
    for n in (1, 3, 4, 1023, 4096, 1000003):
        x = torch.randn(n + 5, generator=gen).to(DEV)
        for off in (0, 1, 4):                                  # 16-byte aligned, 4- and 16-byte offsets
            src = x[off:off + n]
            dst = torch.full((n + 2,), 7.0, device=DEV)
            ops.copy_into(dst[1:1 + n] if off == 1 else dst[:n], src.contiguous())
            got = dst[1:1 + n] if off == 1 else dst[:n]
            assert torch.equal(got, src)
            assert float(dst[-1]) == 7.0                       # nothing past the end
            z = torch.full((n + 2,), 3.0, device=DEV)
            ops.fill_zero(z[1:1 + n])
            assert float(z[0]) == 3.0 and float(z[-1]) == 3.0 and float(z[1:1 + n].abs().max()) == 0.0
    q = torch.randn(128, 513, generator=gen).to(DEV)
    assert torch.equal(ops.clone(q), q)
    zz = ops.zeros(5, 7, 9, dtype=torch.bfloat16, device=DEV)
    assert zz.shape == (5, 7, 9) and zz.dtype == torch.bfloat16 and float(zz.float().abs().max()) == 0.0
    for n in (8, 24, 1000, 4099):
        b = torch.randn(n, generator=gen).to(DEV).to(torch.bfloat16)
        assert torch.equal(ops.cast_f32(b), b.float())
        f = torch.randn(n - n % 8, generator=gen).to(DEV)
        assert torch.equal(ops.cast_to(f, torch.bfloat16), f.to(torch.bfloat16))
    src = torch.randn(64, 7, 8, 4, generator=gen).to(DEV)
    dst0 = torch.randn(64, 7, 7, 3, generator=gen).to(DEV)
    dst = dst0.clone()
    ops.unpad_add(src, dst, 64, 7, 7, 3, 8, 4)
    assert torch.equal(dst, dst0 + src[:, :, :7, :3])
    one = torch.zeros(1, device=DEV)
    assert ops.ones_like_cached(one) is ops.ones_like_cached(one) and float(ops.ones_like_cached(one)) == 1.0


@pytest.mark.parametrize('rows', [4, 8])
def test_conv3x3_wave_kernel_against_ring_kernel(rows):
    """csrc/conv3x3_wave.hip (round 6: one wave per patch, weights resident in LDS, register epilogue) on the 64 -> 64
    3x3 layers of the first stage: the stored bf16 output — forward with fused statistics, the key encoder's affine +
    ReLU form, the data gradient with the BatchNorm-backward epilogue — equals the ring kernel's BIT FOR BIT (same taps,
    same channel order, fp32 accumulation in the MFMA); the statistics slabs are summed in another fixed order: equal
    column totals to fp32 rounding, identical from run to run."""
    from passl_amd.hip import lib as L
    lib = L.load()
    gen = torch.Generator().manual_seed(64)
    cin = cout = 64
    try:
        for H, N in ((56, 3), (8, 5), (24, 2)):
            g = P.ConvGeom(cin, cout, 3, 1, 1)
            fd = P.fwd_desc(g, N, H, H)
            dds, skipped = P.dgrad_plan(g, N, H, H)
            assert len(dds) == 1 and not skipped
            packer = WeightPacker()
            for d in [fd] + dds:
                packer.add(0, cout, 3, 3, cin, d.pack)
            packer.build(DEV, torch.bfloat16).run((torch.randn(cout * 9 * cin, generator=gen) * 0.05).to(DEV))
            x = torch.randn(N, H, H, cin, generator=gen).to(DEV).to(torch.bfloat16)
            dy = torch.randn(N, H, H, cout, generator=gen).to(DEV).to(torch.bfloat16)
            yb = torch.randn(N, H, H, cin, generator=gen).to(DEV).to(torch.bfloat16)
            sc, sf = torch.rand(cout, generator=gen).to(DEV) + 0.5, torch.randn(cout, generator=gen).to(DEV) * 0.3
            slab, tiles = ops.conv_stats_buffer(fd, DEV)
            d = dds[0]
            td = ops.conv_tiles(d)
            part = torch.zeros(ops.bn_partial_floats(td, cin, False), device=DEV)
            bnb = dict(y=yb, mask=None, mean=torch.randn(cin, generator=gen).to(DEV) * 0.1,
                       invstd=torch.rand(cin, generator=gen).to(DEV) + 0.5, scale=sc, shift=sf, relu=2, partial=part,
                       tile_off=0)
            got = {}
            for wave in (0, 1, 1):
                assert lib.passl_hip_set_option(b'conv3x3_wave', wave) == 0
                assert lib.passl_hip_set_option(b'conv3x3_wave_rows', rows) == 0
                y = torch.zeros(N, H, H, cout, device=DEV, dtype=torch.bfloat16)
                yk = torch.zeros_like(y)
                dx = torch.zeros(N, H, H, cin, device=DEV, dtype=torch.bfloat16)
                slab.fill_(float('nan')); part.fill_(float('nan'))
                ops.conv_igemm(fd, x, packer.view(fd.pack, cout), y, stats=slab)
                k1 = lib.passl_hip_last_igemm_kernel()
                ops.conv_igemm(fd, x, packer.view(fd.pack, cout), yk, scale=sc, shift=sf, relu=True)
                k2 = lib.passl_hip_last_igemm_kernel()
                ops.conv_igemm(d, dy, packer.view(d.pack, cin), dx, bnb=bnb)
                k3 = lib.passl_hip_last_igemm_kernel()
                assert (k1, k2, k3) == ((4, 4, 4) if wave else (1, 1, 1)), (k1, k2, k3)
                # column totals of the slabs in fp64: sum (v - s) + n s, sum g, sum g xhat
                sl = slab[:tiles * cout * 3].double()
                sums = sl[:tiles * cout * 2].view(tiles, cout, 2)
                shifts = sl[tiles * cout * 2:].view(tiles, cout)
                nrow = torch.full((tiles, 1), 128.0, device=DEV, dtype=torch.float64)
                nrow[-1] = N * H * H - 128 * (tiles - 1)
                tot1 = (sums[:, :, 0] + nrow * shifts).sum(0)
                tot2 = (sums[:, :, 1] + 2 * shifts * sums[:, :, 0] + nrow * shifts * shifts).sum(0)
                pt = part[:td * cin * 2].double().view(td, cin, 2).sum(0)
                got.setdefault(wave, []).append([t.clone() for t in (y, yk, dx, tot1, tot2, pt[:, 0], pt[:, 1],
                                                                     slab[:tiles * cout * 3], part[:td * cin * 2])])
            ring, w1, w2 = got[0][0], got[1][0], got[1][1]
            for i in range(3):                                   # the three stored tensors: bit for bit
                assert torch.equal(_bits(ring[i]), _bits(w1[i])), (H, N, i)
            yd = ring[0].double().reshape(-1, cout)
            assert float((w1[3] - yd.sum(0)).abs().max()) < 1e-3 * float(yd.abs().sum(0).max())     # sums of the STORED values
            assert float((w1[4] - (yd * yd).sum(0)).abs().max()) < 1e-3 * float((yd * yd).sum(0).max())
            for i in (3, 4, 5, 6):                               # ... and the ring kernel's slabs say the same
                scale = float(ring[i].abs().max()) + 1e-6
                assert float((ring[i] - w1[i]).abs().max()) < 2e-5 * scale + 1e-4, (H, N, i)
            for i in range(9):                                   # second run of the wave kernel: identical bits everywhere
                assert not torch.isnan(w1[i].float()).any()
                assert torch.equal(_bits(w1[i]), _bits(w2[i])), (H, N, i)
    finally:
        assert lib.passl_hip_set_option(b'conv3x3_wave', 1) == 0
        assert lib.passl_hip_set_option(b'conv3x3_wave_rows', 4) == 0
