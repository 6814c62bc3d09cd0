"""Tensor-level wrappers over the C ABI (no autograd here; see functional.py).

torch is used for device memory and the current stream only.  Every function launches HIP
kernels from libpassl_hip.so; host tensors are refused (lib.ptr raises).
"""
import ctypes as C

import torch

from . import config
from . import lib as L
from . import plan as P


def _lib():
    return L.load()


# ------------------------------------------------------------------ shared fp32 workspace
class _Workspace:
    """ONE grow-only fp32 scratch buffer per device for the fixed-order (atomic-free) reductions:
    split-M weight-gradient slabs, InfoNCE dq slabs, column-sum partials.  Every user launches its
    producer and its reduce kernel back to back on the current stream, so stream order makes sharing
    safe; the buffer is never read across ops.  Keyed by stream as well: the weight gradients run on the
    side stream (hip/streams.py) with their own scratch."""

    def __init__(self):
        self._bufs = {}

    def get(self, n_floats, device):
        # one buffer per (device, stream): users on different streams must not share scratch
        key = (device.type, device.index, L.stream() if device.type == 'cuda' else 0)     # raw handle: ~0.3 us
        buf = self._bufs.get(key)
        if buf is None or buf.numel() < n_floats:
            # grow-only with a 4 MB floor (most users need a few KB; the largest weight-gradient slabs ask for more)
            buf = torch.empty(max(int(n_floats), 1 << 20), dtype=torch.float32, device=device)
            self._bufs[key] = buf
        return buf

    def pin(self):
        """References to every scratch buffer that exists right now, for somebody whose recorded launches carry raw
        pointers into them (hip/replay.py): ``get`` REPLACES a buffer when a later call asks for more, and the old one
        must then outlive its last recorded user.  Dropping the returned list is the unpin.  (The library's own
        finalize-counter slices, csrc/bn.hip, live as long as the library: recorded launches hold those too, and a
        plan and an eager step must not run concurrently on one device for that reason.)"""
        return list(self._bufs.values())


workspace = _Workspace()


# ------------------------------------------------------------------ fills / copies / casts
# Library kernels for what would otherwise be the step's last ATen launches: a step plan (hip/replay.py) replays
# the launches of THIS library, so a training step must not depend on a kernel torch launches.
def fill_zero(t):
    """t[...] = 0 for a contiguous device tensor."""
    assert t.is_contiguous()
    if t.numel():
        L.check(_lib().passl_hip_fill_zero(L.ptr(t), t.numel() * t.element_size(), L.stream()), 'fill_zero')
    return t


def zeros(*shape, dtype=None, device=None):
    return fill_zero(torch.empty(*shape, dtype=dtype, device=device))


def zeros_like(x):
    return fill_zero(torch.empty(x.shape, dtype=x.dtype, device=x.device))


def copy_into(dst, src):
    """dst[...] = src[...]: same dtype and element count, both contiguous, no overlap."""
    assert dst.is_contiguous() and src.is_contiguous() and dst.dtype == src.dtype and dst.numel() == src.numel()
    if dst.numel():
        L.check(_lib().passl_hip_copy_bytes(L.ptr(dst), L.ptr(src), dst.numel() * dst.element_size(), L.stream()),
                'copy_bytes')
    return dst


def clone(x):
    return copy_into(torch.empty(x.shape, dtype=x.dtype, device=x.device), x.contiguous())


def cast_f32(x):
    """bf16 -> fp32 copy (fp32 input is returned as is)."""
    if x.dtype == torch.float32:
        return x
    assert x.dtype == torch.bfloat16
    x = x.contiguous()
    y = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    L.check(_lib().passl_hip_cast_bf16_to_f32(L.ptr(x), L.ptr(y), x.numel(), L.stream()), 'cast_bf16_to_f32')
    return y


def cast_to(x, dtype):
    """fp32 <-> bf16 copy through the library's cast kernels (identity when the dtype already matches)."""
    if x.dtype == dtype:
        return x
    if dtype == torch.float32:
        return cast_f32(x)
    assert dtype == torch.bfloat16 and x.dtype == torch.float32
    x = x.contiguous()
    y = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    cast_bf16(x.view(-1), y.view(-1))
    return y


def unpad_add(src, dst, rows, R, dst_S, dst_C, src_S, src_C):
    """dst[row][r][s][c] += src[row][r][s][c] over the dense extents of a padded block (the stem filter's gradient)."""
    L.check(_lib().passl_hip_unpad_add(L.ptr(src), L.ptr(dst), rows, R, dst_S, dst_C, src_S, src_C, L.stream()),
            'unpad_add')


def add_into(dst, src):
    """dst += src for contiguous fp32 tensors of the same size (the fixed-order slab adder with one slab)."""
    assert dst.dtype == torch.float32 and src.dtype == torch.float32 and dst.numel() == src.numel()
    assert dst.is_contiguous() and src.is_contiguous()
    L.check(_lib().passl_hip_slab_reduce(L.ptr(src), L.ptr(dst), dst.numel(), 1, 1, L.stream()), 'slab_reduce')
    return dst


_ones = {}


def ones_like_cached(t):
    """A resident all-ones tensor shaped like t (the root gradient of ``loss.backward()``: autograd would otherwise
    launch a fill kernel per step)."""
    key = (t.device, t.dtype, tuple(t.shape))
    o = _ones.get(key)
    if o is None:
        o = _ones[key] = torch.ones(t.shape, dtype=t.dtype, device=t.device)
    return o


# ------------------------------------------------------------------ convolution
_desc_cache = {}


def _conv_struct(d: P.Desc):
    key = id(d)
    hit = _desc_cache.get(key)
    if hit is not None and hit[0] is d:
        return hit[1]
    s = L.ConvDesc()
    for f in ('N', 'OP', 'OQ', 'NCOLS', 'R', 'S', 'C', 'IH', 'IW', 'sh', 'sw', 'ph', 'pw',
              'a_sn', 'a_sh', 'a_sw', 'y_sn', 'y_sh', 'y_sw'):
        setattr(s, f, getattr(d, f))
    s.stats = None
    s.stats_tiles = 0
    _desc_cache[key] = (d, s)
    return s


def conv_tiles(d: P.Desc):
    """Number of 128-row output tiles of a launch = slab rows of its fused statistics."""
    return (d.N * d.OP * d.OQ + 127) // 128


def conv_igemm(d: P.Desc, a, b, y, scale=None, shift=None, residual=None, relu=False,
               out_f32=False, stats=None, bnb=None):
    """Launch one implicit-GEMM conv described by `d` (plan.Desc).  `a` activation tensor,
    `b` packed weights [NCOLS, R*S*C] (same dtype as a), `y` output tensor (written in place at
    d.y_off with d's strides).  `stats`: fp32 slab from conv_stats_buffer (fused forward BatchNorm
    statistics of the stored output, bf16 only).  `bnb`: dict(y, mask, mean, invstd, scale, shift,
    relu, partial, tile_off) — BatchNorm-backward statistics fused into this data-gradient launch
    (include/passl_hip.h: passl_conv_desc.bnb_*)."""
    s = _conv_struct(d)
    esz = 4 if out_f32 else y.element_size()
    s.a = L.ptr(a)
    s.b = L.ptr(b)
    s.y = L.ptr(y) + d.y_off * esz
    s.scale = L.ptr(scale)
    s.shift = L.ptr(shift)
    s.residual = (L.ptr(residual) + d.y_off * esz) if residual is not None else None
    s.relu = 1 if relu else 0
    s.dtype = L.dt(a)
    s.out_f32 = 1 if out_f32 else 0
    s.stats = L.ptr(stats)
    s.stats_tiles = conv_tiles(d) if stats is not None else 0
    if bnb is not None:
        s.bnb_y = L.ptr(bnb['y']) + d.y_off * esz
        # the bit mask is indexed like the dense tensor: shift it with the sub-lattice origin (y_off % 8 == 0)
        mask = bnb.get('mask')
        s.bnb_mask = (L.ptr(mask) + (d.y_off >> 3)) if mask is not None else None
        s.bnb_mean, s.bnb_invstd = L.ptr(bnb['mean']), L.ptr(bnb['invstd'])
        s.bnb_scale, s.bnb_shift = L.ptr(bnb.get('scale')), L.ptr(bnb.get('shift'))
        s.bnb_partial = L.ptr(bnb['partial'])
        s.bnb_relu, s.bnb_tile_off = int(bnb['relu']), int(bnb['tile_off'])
        if bnb.get('partial2') is not None:          # a second BatchNorm behind the same gradient (bnb2_*)
            s.bnb2_y = L.ptr(bnb['y2']) + d.y_off * esz
            s.bnb2_mean, s.bnb2_invstd = L.ptr(bnb['mean2']), L.ptr(bnb['invstd2'])
            s.bnb2_partial = L.ptr(bnb['partial2'])
        else:
            s.bnb2_y = s.bnb2_mean = s.bnb2_invstd = s.bnb2_partial = None
    else:
        s.bnb_y = s.bnb_mask = s.bnb_mean = s.bnb_invstd = s.bnb_scale = s.bnb_shift = None
        s.bnb_partial = None
        s.bnb_relu = s.bnb_tile_off = 0
        s.bnb2_y = s.bnb2_mean = s.bnb2_invstd = s.bnb2_partial = None
    rc = _lib().passl_hip_conv_igemm(C.byref(s), L.stream())
    if rc == L.EUNSUPPORTED and bnb is not None and bnb.get('partial2') is not None:
        # include/passl_hip.h (bnb2_*): the two-BatchNorm instantiations cover fewer launches than the Python-side test
        # can know (K-tile thresholds, kernel selection options, operand strides): run this launch without the second
        # layer — its own backward then does its reduce pass — and tell the caller through bnb['partial2_done']
        s.bnb2_y = s.bnb2_mean = s.bnb2_invstd = s.bnb2_partial = None
        bnb['partial2_done'] = False
        rc = _lib().passl_hip_conv_igemm(C.byref(s), L.stream())
    elif bnb is not None and bnb.get('partial2') is not None:
        bnb['partial2_done'] = True
    L.check(rc, 'conv_igemm')
    return y


_wdesc_cache = {}


def wgrad_slices(d: P.Desc, dtype):
    """Number of reduction slices of a weight-gradient launch (host logic, no GPU): the product kernels' heuristic,
    or — only with the EXPERIMENTAL spatially tiled kernel switched on (config.wgrad_halo) and for the launches it
    takes — the count that fits ITS grid."""
    halo = config.wgrad_halo()
    if (halo and dtype == torch.bfloat16 and d.R == 3 and d.S == 3 and d.sh == 1 and d.sw == 1 and d.ph == 1 and
            d.pw == 1 and d.IH == d.OP and d.IW == d.OQ and (halo == 2 or (d.IH % 8 == 0 and d.IW % 8 == 0))):
        return P.wgrad_halo_splits(d.N, d.IH, d.IW, d.NCOLS, d.C)
    M = d.N * d.OP * d.OQ
    bkm = 64 if dtype == torch.bfloat16 else 32
    esz = 2 if dtype == torch.bfloat16 else 4
    return P.wgrad_splits(M, d.NCOLS, d.R * d.S * d.C, bkm, target_blocks=P.wgrad_target_blocks(d.IH * d.IW > 1),
                          row_bytes=max(d.NCOLS, d.C * d.sh * d.sw) * esz)


def conv_wgrad(d: P.Desc, a, dy, dw, splits=None):
    """dw[NCOLS, R*S*C] (fp32) += dy^T . gather(a)."""
    key = id(d)
    hit = _wdesc_cache.get(key)
    if hit is not None and hit[0] is d:
        s = hit[1]
    else:
        s = L.WgradDesc()
        for f in ('N', 'OP', 'OQ', 'NCOLS', 'R', 'S', 'C', 'IH', 'IW', 'sh', 'sw', 'ph', 'pw',
                  'a_sn', 'a_sh', 'a_sw'):
            setattr(s, f, getattr(d, f))
        _wdesc_cache[key] = (d, s)
    s.a = L.ptr(a)
    s.dy = L.ptr(dy)
    s.dw = L.ptr(dw)
    s.dy_ld = d.NCOLS
    s.dtype = L.dt(a)
    s.splits = splits or wgrad_slices(d, a.dtype)
    # split-M partial tiles go to slabs of the shared workspace and are added in slice order
    need = s.splits * d.NCOLS * d.R * d.S * d.C
    ws = workspace.get(need, dw.device) if s.splits > 1 else None
    s.ws = L.ptr(ws)
    s.ws_floats = ws.numel() if ws is not None else 0
    L.check(_lib().passl_hip_conv_wgrad(C.byref(s), L.stream()), 'conv_wgrad')
    return dw


# ------------------------------------------------------------------ batch norm
def _bn_blocks(M, Cch, cap=1024):
    lanes = max(1, 256 // max(Cch // 8, 1))
    return int(max(1, min(cap, -(-M // (16 * lanes)))))


def conv_stats_buffer(d: P.Desc, device):
    """Slab for the conv epilogue's fused BN statistics of launch `d`: [tiles][C][2] shifted sums
    followed by [tiles][C] shifts (include/passl_hip.h: passl_conv_desc.stats).  Fully written by the
    kernel: no zeroing, no atomics.  Returns (tensor, tiles)."""
    t = conv_tiles(d)
    return torch.empty(bn_partial_floats(t, d.NCOLS, True), dtype=torch.float32, device=device), t


def bn_partial_floats(nblocks, Cch, shifted):
    """Floats of a BatchNorm partial buffer: slab + the finalize kernels' segment scratch."""
    return nblocks * Cch * (3 if shifted else 2) + 16 * Cch * 4


def _gather_doubles(t):
    """all_gather of a small fp64 device tensor in rank order -> [world, *t.shape]."""
    import torch.distributed as dist
    world = dist.get_world_size()
    out = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, t)            # (concatenation along dim 0: the form every backend accepts)
    return out.view((world,) + tuple(t.shape))


def bn_train_fwd(x, gamma, beta, rmean, rvar, residual=None, relu=True, momentum=0.9, eps=1e-5,
                 partial=None, want_mask=False, sync=False, apply=True):
    """x: [..., C] NHWC rows.  Returns z, stats[4,C] (mean, invstd, scale, shift), relu bit mask
    (or None); updates rmean/rvar in place.  `partial` = (slab, tiles) of fused statistics a conv
    epilogue already wrote (conv_stats_buffer); without it a stats pass over x is launched.
    `apply=False`: statistics only -> (None, stats, None) (the caller applies them: bn_relu_maxpool_fwd).
    `sync`: statistics over the batches of every rank (SyncBatchNorm): this rank's slab is folded to fp64
    moments, the moments are all-gathered and combined in rank order (csrc/bn.hip)."""
    Cch = x.shape[-1]
    M = x.numel() // Cch
    dev = x.device
    stats = torch.empty(4, Cch, dtype=torch.float32, device=dev)   # mean, invstd, scale, shift
    lib, st, dtc = _lib(), L.stream(), L.dt(x)
    if partial is None:
        nb = _bn_blocks(M, Cch)
        rpb = -(-M // nb)
        nb = -(-M // rpb)                   # every slab holds rows
        partial = torch.empty(bn_partial_floats(nb, Cch, True), dtype=torch.float32, device=dev)
        L.check(lib.passl_hip_bn_stats(L.ptr(x), L.ptr(partial), M, Cch, nb, dtc, st), 'bn_stats')
    else:
        partial, nb = partial
        rpb = 128
    if sync:
        mom = torch.empty(3, Cch, dtype=torch.float64, device=dev)
        L.check(lib.passl_hip_bn_moments(L.ptr(partial), nb, M, Cch, rpb, L.ptr(mom), st), 'bn_moments')
        mom_all = _gather_doubles(mom)
        L.check(lib.passl_hip_bn_finalize_moments(L.ptr(mom_all), mom_all.shape[0], Cch, L.ptr(gamma), L.ptr(beta),
                                                  L.ptr(rmean), L.ptr(rvar), momentum, eps, L.ptr(stats[0]),
                                                  L.ptr(stats[1]), L.ptr(stats[2]), L.ptr(stats[3]), L.stream()),
                'bn_finalize_moments')
    else:
        L.check(lib.passl_hip_bn_finalize(L.ptr(partial), nb, M, Cch, rpb, L.ptr(gamma), L.ptr(beta),
                                          L.ptr(rmean), L.ptr(rvar), momentum, eps,
                                          L.ptr(stats[0]), L.ptr(stats[1]), L.ptr(stats[2]),
                                          L.ptr(stats[3]), st), 'bn_finalize')
    if not apply:
        return None, stats, None
    z = torch.empty_like(x)
    mask = torch.empty(x.numel() // 8, dtype=torch.uint8, device=dev) if (want_mask and relu) \
        else None
    L.check(lib.passl_hip_bn_apply(L.ptr(x), L.ptr(stats[2]), L.ptr(stats[3]), L.ptr(residual),
                                   L.ptr(z), L.ptr(mask), M, Cch, 1 if relu else 0, dtc, L.stream()),
            'bn_apply')
    return z, stats, mask


def bn_apply(x, scale, shift, residual=None, relu=False):
    """z = relu?(x*scale + shift + residual) with given per-channel affine (inference BN)."""
    Cch = x.shape[-1]
    M = x.numel() // Cch
    z = torch.empty_like(x)
    L.check(_lib().passl_hip_bn_apply(L.ptr(x), L.ptr(scale), L.ptr(shift), L.ptr(residual),
                                      L.ptr(z), None, M, Cch, 1 if relu else 0, L.dt(x),
                                      L.stream()), 'bn_apply')
    return z


def bn_bwd(dz, z, x, gamma, mean, invstd, dgamma, dbeta, relu=True, want_dres=False,
           scale=None, shift=None, fused=None, sync=False):
    """Returns dx (and dres).  dgamma/dbeta (fp32 [C]) are accumulated into.
    `relu`: False/0 none, True/1 mask = z > 0, 2 mask recomputed from x*scale+shift (z unused),
    3 `z` is the bit mask written by the forward's bn_apply.
    `fused` = (slab, tiles): the data-gradient launch that produced `dz` already masked it and wrote
    the (sum g, sum g*xhat) slab (conv_igemm(bnb=...)): no reduce pass, no mask in the apply pass,
    and the residual-branch gradient IS dz (returned as dres without a copy)."""
    Cch = x.shape[-1]
    M = x.numel() // Cch
    dev = x.device
    coef = torch.empty(3 * Cch, dtype=torch.float32, device=dev)
    lib, st, dtc = _lib(), L.stream(), L.dt(x)
    r = int(relu)
    zp = L.ptr(z) if r in (1, 3) else None
    if fused is not None:
        partial, nb = fused
        r, zp = 0, None
    else:
        # 768 = 3 resident workgroups per CU x 256 CUs: the backward-reduce kernel needs 136 VGPRs (3 waves per SIMD), and
        # a 1024-block launch ran a second, one-third-occupied round (the stem's reduce: 822 MB in 292 us = 2.8 TB/s)
        nb = _bn_blocks(M, Cch, cap=768)
        partial = torch.empty(bn_partial_floats(nb, Cch, False), dtype=torch.float32, device=dev)
        L.check(lib.passl_hip_bn_bwd_reduce(L.ptr(dz), zp, L.ptr(x), L.ptr(mean), L.ptr(invstd),
                                            L.ptr(scale), L.ptr(shift), L.ptr(partial), M, Cch, nb, r,
                                            dtc, st), 'bn_bwd_reduce')
    if sync:         # SyncBatchNorm: {sum g, sum g*xhat} of every rank, row count of the global batch
        import torch.distributed as dist
        sums = torch.empty(2, Cch, dtype=torch.float64, device=dev)
        L.check(lib.passl_hip_bn_bwd_sums(L.ptr(partial), nb, M, Cch, L.ptr(sums), st), 'bn_bwd_sums')
        sums_all = _gather_doubles(sums)
        L.check(lib.passl_hip_bn_bwd_finalize_sums(L.ptr(sums_all), sums_all.shape[0], dist.get_rank(),
                                                   M * sums_all.shape[0], Cch, L.ptr(gamma), L.ptr(mean),
                                                   L.ptr(invstd), L.ptr(dgamma), L.ptr(dbeta), L.ptr(coef),
                                                   L.stream()), 'bn_bwd_finalize_sums')
    else:
        L.check(lib.passl_hip_bn_bwd_finalize(L.ptr(partial), nb, M, Cch, L.ptr(gamma), L.ptr(mean),
                                              L.ptr(invstd), L.ptr(dgamma), L.ptr(dbeta), L.ptr(coef),
                                              st), 'bn_bwd_finalize')
    dx = torch.empty_like(x)
    if fused is not None:
        dres_out, dres = None, (dz if want_dres else None)
    else:
        dres_out = dres = torch.empty_like(x) if want_dres else None
    L.check(lib.passl_hip_bn_bwd_apply(L.ptr(dz), zp, L.ptr(x), L.ptr(coef), L.ptr(scale),
                                       L.ptr(shift), L.ptr(dx), L.ptr(dres_out), M, Cch, r, dtc, L.stream()),
            'bn_bwd_apply')
    return dx, dres


# ------------------------------------------------------------------ pooling / layout
def nchw_to_nhwc_pad(x, pad, Wp, Cp, dtype):
    N, Cc, H, W = x.shape
    y = torch.empty(N, H + 2 * pad, Wp, Cp, dtype=dtype, device=x.device)
    L.check(_lib().passl_hip_nchw_to_nhwc_pad(L.ptr(x), L.ptr(y), N, Cc, H, W, pad, Wp, Cp,
                                              L.dt(dtype), L.stream()), 'nchw_to_nhwc_pad')
    return y


def maxpool_fwd(x):
    N, H, W, Cc = x.shape
    P_, Q_ = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    y = torch.empty(N, P_, Q_, Cc, dtype=x.dtype, device=x.device)
    idx = torch.empty(N, P_, Q_, Cc, dtype=torch.uint8, device=x.device)
    L.check(_lib().passl_hip_maxpool3x3s2_fwd(L.ptr(x), L.ptr(y), L.ptr(idx), N, H, W, Cc,
                                              L.dt(x), L.stream()), 'maxpool_fwd')
    return y, idx


def maxpool_bwd(dy, idx, H, W):
    N, _, _, Cc = dy.shape
    dx = torch.empty(N, H, W, Cc, dtype=dy.dtype, device=dy.device)
    L.check(_lib().passl_hip_maxpool3x3s2_bwd(L.ptr(dy), L.ptr(idx), L.ptr(dx), N, H, W, Cc,
                                              L.dt(dy), L.stream()), 'maxpool_bwd')
    return dx


def bn_relu_maxpool_supported(x):
    """The fused stem pass (csrc/stem_pool.hip) takes this BatchNorm input: [N,H,W,C] with C/8 dividing 256."""
    return x.dim() == 4 and _lib().passl_hip_bn_relu_maxpool_blocks(*x.shape) > 0


def bn_relu_maxpool_fwd(x, stats):
    """max-pool(relu(x * scale + shift)) in one pass; stats[4,C] from bn_train_fwd(..., apply=False) -> y, idx."""
    N, H, W, Cc = x.shape
    P_, Q_ = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    y = torch.empty(N, P_, Q_, Cc, dtype=x.dtype, device=x.device)
    idx = torch.empty(N, P_, Q_, Cc, dtype=torch.uint8, device=x.device)
    L.check(_lib().passl_hip_bn_relu_maxpool_fwd(L.ptr(x), L.ptr(stats[2]), L.ptr(stats[3]), L.ptr(y), L.ptr(idx),
                                                 N, H, W, Cc, L.dt(x), L.stream()), 'bn_relu_maxpool_fwd')
    return y, idx


def bn_relu_maxpool_bwd(dy, idx, x, gamma, stats, dgamma, dbeta, parts=1, on_part=None):
    """Backward of bn_relu_maxpool_fwd w.r.t. x; dgamma / dbeta (fp32 [C]) are accumulated into.
    parts > 1: the apply pass runs over that many pieces of the batch; on_part(dx, n0, n1, i, parts) is called after the launch
    that wrote dx[n0:n1], piece i of `parts` (the caller hands the piece to whoever consumes it: the stem's weight gradient)."""
    N, H, W, Cc = x.shape
    lib, st, dtc, dev = _lib(), L.stream(), L.dt(x), x.device
    nb = lib.passl_hip_bn_relu_maxpool_blocks(N, H, W, Cc)
    partial = torch.empty(bn_partial_floats(nb, Cc, False), dtype=torch.float32, device=dev)
    coef = torch.empty(3 * Cc, dtype=torch.float32, device=dev)
    L.check(lib.passl_hip_bn_relu_maxpool_bwd_reduce(L.ptr(dy), L.ptr(idx), L.ptr(x), L.ptr(stats[0]), L.ptr(stats[1]),
                                                     L.ptr(stats[2]), L.ptr(stats[3]), L.ptr(partial), nb, N, H, W, Cc,
                                                     dtc, st), 'bn_relu_maxpool_bwd_reduce')
    L.check(lib.passl_hip_bn_bwd_finalize(L.ptr(partial), nb, N * H * W, Cc, L.ptr(gamma), L.ptr(stats[0]),
                                          L.ptr(stats[1]), L.ptr(dgamma), L.ptr(dbeta), L.ptr(coef), st),
            'bn_bwd_finalize')
    dx = torch.empty_like(x)
    parts = max(1, min(int(parts), N))
    for i in range(parts):
        n0, n1 = N * i // parts, N * (i + 1) // parts
        L.check(lib.passl_hip_bn_relu_maxpool_bwd_apply(L.ptr(dy[n0:n1]), L.ptr(idx[n0:n1]), L.ptr(x[n0:n1]), L.ptr(coef),
                                                        L.ptr(stats[2]), L.ptr(stats[3]), L.ptr(dx[n0:n1]), n1 - n0, H, W,
                                                        Cc, dtc, st), 'bn_relu_maxpool_bwd_apply')
        if on_part is not None:
            on_part(dx, n0, n1, i, parts)
    return dx


def avgpool_fwd(x):
    N, H, W, Cc = x.shape
    y = torch.empty(N, Cc, dtype=x.dtype, device=x.device)
    L.check(_lib().passl_hip_avgpool_fwd(L.ptr(x), L.ptr(y), N, H * W, Cc, L.dt(x), L.stream()),
            'avgpool_fwd')
    return y


def avgpool_bwd(dy, H, W):
    N, Cc = dy.shape
    dx = torch.empty(N, H, W, Cc, dtype=dy.dtype, device=dy.device)
    L.check(_lib().passl_hip_avgpool_bwd(L.ptr(dy), L.ptr(dx), N, H * W, Cc, L.dt(dy), L.stream()),
            'avgpool_bwd')
    return dx


def relu_bwd(dy, y):
    dx = torch.empty_like(dy)
    L.check(_lib().passl_hip_relu_bwd(L.ptr(dy), L.ptr(y), L.ptr(dx), dy.numel(), L.dt(dy),
                                      L.stream()), 'relu_bwd')
    return dx


def colsum_into(x, out, accumulate=False):
    """out[C] (fp32) = (accumulate: +=) column sums of x [M, C]."""
    M, Cc = x.shape
    fn = _lib().passl_hip_colsum_acc if accumulate else _lib().passl_hip_colsum
    ws = workspace.get(1024 * Cc, x.device)
    L.check(fn(L.ptr(x), L.ptr(out), M, Cc, L.dt(x), L.ptr(ws), ws.numel(), L.stream()), 'colsum')
    return out


# ------------------------------------------------------------------ contrastive head
def l2norm_fwd(x, eps=1e-12):
    N, Dd = x.shape
    y = torch.empty_like(x)
    norm = torch.empty(N, dtype=torch.float32, device=x.device)
    L.check(_lib().passl_hip_l2norm_fwd(L.ptr(x), L.ptr(y), L.ptr(norm), N, Dd, eps, L.stream()),
            'l2norm_fwd')
    return y, norm


def l2norm_bwd(dy, y, norm, out_dtype):
    N, Dd = y.shape
    dx = torch.empty(N, Dd, dtype=out_dtype, device=y.device)
    L.check(_lib().passl_hip_l2norm_bwd(L.ptr(dy), L.ptr(y), L.ptr(norm), L.ptr(dx), N, Dd,
                                        L.dt(out_dtype), L.stream()), 'l2norm_bwd')
    return dx


def cosine_loss_fwd(a, b, eps=1e-8):
    """-> loss [1] = -mean_i cos(a_i, b_i), stats [N,4] (for the backward)."""
    N, Dd = a.shape
    stats = torch.empty(N, 4, dtype=torch.float32, device=a.device)
    loss = torch.empty(1, dtype=torch.float32, device=a.device)
    L.check(_lib().passl_hip_cosine_loss_fwd(L.ptr(a), L.ptr(b), N, Dd, eps, L.ptr(stats), L.ptr(loss), L.stream()),
            'cosine_loss_fwd')
    return loss, stats


def cosine_loss_bwd(a, b, stats, gloss):
    N, Dd = a.shape
    da = torch.empty_like(a)
    L.check(_lib().passl_hip_cosine_loss_bwd(L.ptr(a), L.ptr(b), L.ptr(stats), L.ptr(gloss), N, Dd, L.ptr(da),
                                             L.stream()), 'cosine_loss_bwd')
    return da


def infonce_fwd(q, k, queue, T, want_logits=False):
    """Returns out[3] (loss, acc1, acc5), row_lse[N], logits[N,K+1] or None."""
    N, Dd = q.shape
    K = queue.shape[1]
    lib = _lib()
    ws = torch.empty(max(lib.passl_hip_infonce_workspace_bytes(N, K) // 4, 4),
                     dtype=torch.float32, device=q.device)
    out = torch.empty(3, dtype=torch.float32, device=q.device)
    lse = torch.empty(N, dtype=torch.float32, device=q.device)
    logits = torch.empty(N, K + 1, dtype=torch.float32, device=q.device) if want_logits else None
    L.check(lib.passl_hip_infonce_fwd(L.ptr(q), L.ptr(k), L.ptr(queue), N, Dd, K, T, L.ptr(out),
                                      L.ptr(lse), L.ptr(logits), L.ptr(ws), L.stream()),
            'infonce_fwd')
    return out, lse, logits


def infonce_bwd(q, k, queue, lse, gscale, T):
    N, Dd = q.shape
    K = queue.shape[1]
    dq = torch.empty_like(q)
    lib = _lib()
    ws = workspace.get(lib.passl_hip_infonce_bwd_workspace_bytes(N, K) // 4, q.device)
    L.check(lib.passl_hip_infonce_bwd(L.ptr(q), L.ptr(k), L.ptr(queue), L.ptr(lse),
                                      L.ptr(gscale), N, Dd, K, T, L.ptr(dq), L.ptr(ws), L.stream()),
            'infonce_bwd')
    return dq


def enqueue(queue, keys, ptr):
    Dd, K = queue.shape
    B = keys.shape[0]
    L.check(_lib().passl_hip_enqueue(L.ptr(queue), L.ptr(keys), Dd, K, int(ptr), B, L.stream()),
            'enqueue')


# ------------------------------------------------------------------ flat buffers
def ema_update(k_flat, q_flat, m, k_lp=None):
    L.check(_lib().passl_hip_ema_update(L.ptr(k_flat), L.ptr(q_flat), L.ptr(k_lp),
                                        k_flat.numel(), m, L.stream()), 'ema_update')


def bn_fold(flat, idx, eps, scale, shift):
    """scale / shift of every inference-form BatchNorm of an encoder (idx = (gamma, beta, mean, var) int64
    index lists into `flat`), one launch."""
    gi, bi, mi, vi = idx
    L.check(_lib().passl_hip_bn_fold(L.ptr(flat), L.ptr(gi), L.ptr(bi), L.ptr(mi), L.ptr(vi), gi.numel(),
                                     float(eps), L.ptr(scale), L.ptr(shift), L.stream()), 'bn_fold')


def momentum_sgd(p, g, v, lr, mu, wd, grad_scale=1.0):
    L.check(_lib().passl_hip_momentum_sgd(L.ptr(p), L.ptr(g), L.ptr(v), p.numel(), lr, mu, wd,
                                          grad_scale, L.stream()), 'momentum_sgd')


def momentum_sgd_dev(p, g, v, hyper, mu, wd, grad_scale=1.0):
    """hyper: device float32 tensor, [0] = learning rate (read when the kernel runs: HIP-graph replays)."""
    L.check(_lib().passl_hip_momentum_sgd_dev(L.ptr(p), L.ptr(g), L.ptr(v), p.numel(), L.ptr(hyper), mu, wd,
                                              grad_scale, L.stream()), 'momentum_sgd_dev')


def enqueue_dev(queue, keys, ptr_dev):
    """ptr_dev: int64[1] device tensor (MoCo's queue_ptr buffer); read and advanced on the device."""
    Dd, K = queue.shape
    L.check(_lib().passl_hip_enqueue_dev(L.ptr(queue), L.ptr(keys), Dd, K, L.ptr(ptr_dev), keys.shape[0],
                                         L.stream()), 'enqueue_dev')


def cast_bf16(src, dst):
    L.check(_lib().passl_hip_cast_f32_to_bf16(L.ptr(src), L.ptr(dst), src.numel(), L.stream()),
            'cast_f32_to_bf16')


def pack_weights(src, dst, jobs_dev, block_job, block_start, n_blocks):
    L.check(_lib().passl_hip_pack_weights(L.ptr(src), L.ptr(dst), L.dt(dst), L.ptr(jobs_dev),
                                          L.ptr(block_job), L.ptr(block_start), n_blocks,
                                          L.stream()), 'pack_weights')


# ------------------------------------------------------------------ SimCLR head / LARS
def ntxent_fwd(a, b, a_all, b_all, row_offset, T, co2_weight=3.0):
    """Returns out[2] (loss, acc1) and rowstats [B, 8] (saved for the backward)."""
    B, Dd = a.shape
    out = torch.empty(2, dtype=torch.float32, device=a.device)
    rowstats = torch.empty(B, 8, dtype=torch.float32, device=a.device)
    L.check(_lib().passl_hip_ntxent_fwd(L.ptr(a), L.ptr(b), L.ptr(a_all), L.ptr(b_all), B,
                                        a_all.shape[0], int(row_offset), Dd, T, co2_weight,
                                        L.ptr(out), L.ptr(rowstats), L.stream()), 'ntxent_fwd')
    return out, rowstats


def ntxent_bwd(a, b, a_all, b_all, rowstats, gscale, row_offset, T, co2_weight=3.0):
    """Returns (da, db, da_all, db_all): row-role and column-role gradients."""
    B, Dd = a.shape
    da, db = zeros_like(a), zeros_like(b)
    da_all, db_all = zeros_like(a_all), zeros_like(b_all)
    L.check(_lib().passl_hip_ntxent_bwd(L.ptr(a), L.ptr(b), L.ptr(a_all), L.ptr(b_all),
                                        L.ptr(rowstats), L.ptr(gscale), B, a_all.shape[0],
                                        int(row_offset), Dd, T, co2_weight, L.ptr(da), L.ptr(db),
                                        L.ptr(da_all), L.ptr(db_all), L.stream()), 'ntxent_bwd')
    return da, db, da_all, db_all


def lars_momentum(p, g, v, table, lr, mu, coeff, eps, grad_scale=1.0):
    """`table` = dict(blk_off int64[nb], blk_len int32[nb], blk_seg int32[nb], seg_wd float[ns],
    norms float[ns + nb, 2] = workspace: squared norms + per-block partial sums) on the device (built
    once by the optimizer)."""
    assert table['norms'].numel() >= 2 * (table['seg_wd'].numel() + table['blk_off'].numel())
    L.check(_lib().passl_hip_lars_momentum(L.ptr(p), L.ptr(g), L.ptr(v), L.ptr(table['blk_off']),
                                           L.ptr(table['blk_len']), L.ptr(table['blk_seg']),
                                           table['blk_off'].numel(), L.ptr(table['seg_wd']),
                                           table['seg_wd'].numel(), L.ptr(table['norms']), lr, mu,
                                           coeff, eps, grad_scale, L.stream()), 'lars_momentum')


def lars_momentum_dev(p, g, v, table, hyper, mu, coeff, eps, grad_scale=1.0):
    """lars_momentum with lr = hyper[0] read on the device."""
    assert table['norms'].numel() >= 2 * (table['seg_wd'].numel() + table['blk_off'].numel())
    L.check(_lib().passl_hip_lars_momentum_dev(L.ptr(p), L.ptr(g), L.ptr(v), L.ptr(table['blk_off']),
                                               L.ptr(table['blk_len']), L.ptr(table['blk_seg']),
                                               table['blk_off'].numel(), L.ptr(table['seg_wd']),
                                               table['seg_wd'].numel(), L.ptr(table['norms']), L.ptr(hyper), mu,
                                               coeff, eps, grad_scale, L.stream()), 'lars_momentum_dev')


def larc_momentum_dev(p, g, v, table, hyper, mu, trust, eps, clip, grad_scale=1.0):
    """MomentumLARC over the flat arena (include/passl_hip.h), lr = hyper[0] on the device; same table as LARS."""
    assert table['norms'].numel() >= 2 * (table['seg_wd'].numel() + table['blk_off'].numel())
    L.check(_lib().passl_hip_larc_momentum_dev(L.ptr(p), L.ptr(g), L.ptr(v), L.ptr(table['blk_off']),
                                               L.ptr(table['blk_len']), L.ptr(table['blk_seg']),
                                               table['blk_off'].numel(), L.ptr(table['seg_wd']),
                                               table['seg_wd'].numel(), L.ptr(table['norms']), L.ptr(hyper), mu,
                                               trust, eps, int(bool(clip)), grad_scale, L.stream()),
            'larc_momentum_dev')


# ------------------------------------------------------------------ ViT / MAE
def layernorm_fwd(x, gamma, beta, eps):
    Cc = x.shape[-1]
    M = x.numel() // Cc
    y = torch.empty_like(x)
    mean = torch.empty(M, dtype=torch.float32, device=x.device)
    rstd = torch.empty(M, dtype=torch.float32, device=x.device)
    L.check(_lib().passl_hip_layernorm_fwd(L.ptr(x), L.ptr(gamma), L.ptr(beta), L.ptr(y), L.ptr(mean),
                                           L.ptr(rstd), M, Cc, eps, L.dt(x), L.stream()), 'layernorm_fwd')
    return y, mean, rstd


def layernorm_bwd(dy, x, gamma, mean, rstd, dgamma, dbeta, dres=None, defer_params=False):
    """dres: gradient of the residual branch that forked off x (added to dx inside the kernel).
    defer_params: only dx now; -> (dx, partials, fold) where ``fold()`` adds the fixed-order parameter sums into
    dgamma / dbeta on whatever stream is current when it is called (hip/nn.py runs it on the side stream: a
    latency-bound launch the input-gradient chain does not wait for)."""
    Cc = x.shape[-1]
    M = x.numel() // Cc
    dx = torch.empty_like(x)
    need = -(-M // 16) * 2 * Cc                              # >= passl_hip_layernorm_bwd_ws_floats(M, C)
    lib = _lib()
    if defer_params:
        ws = torch.empty(need, dtype=torch.float32, device=x.device)      # its own buffer: read later, elsewhere
        L.check(lib.passl_hip_layernorm_bwd(L.ptr(dy), L.ptr(x), L.ptr(gamma), L.ptr(mean), L.ptr(rstd),
                                            L.ptr(dres) if dres is not None else None, L.ptr(dx), None, None, M, Cc,
                                            L.dt(x), L.ptr(ws), ws.numel(), L.stream()), 'layernorm_bwd')

        def fold():
            L.check(lib.passl_hip_layernorm_param_reduce(L.ptr(ws), M, Cc, L.ptr(dgamma), L.ptr(dbeta), L.stream()),
                    'layernorm_param_reduce')
        return dx, ws, fold
    ws = workspace.get(need, x.device)
    L.check(lib.passl_hip_layernorm_bwd(L.ptr(dy), L.ptr(x), L.ptr(gamma), L.ptr(mean), L.ptr(rstd),
                                        L.ptr(dres) if dres is not None else None, L.ptr(dx),
                                        L.ptr(dgamma), L.ptr(dbeta), M, Cc, L.dt(x), L.ptr(ws), ws.numel(),
                                        L.stream()), 'layernorm_bwd')
    return dx


def gelu_fwd(x):
    y = torch.empty_like(x)
    L.check(_lib().passl_hip_gelu_fwd(L.ptr(x), L.ptr(y), x.numel(), L.dt(x), L.stream()), 'gelu_fwd')
    return y


def gelu_bwd(dy, x):
    dx = torch.empty_like(x)
    L.check(_lib().passl_hip_gelu_bwd(L.ptr(dy), L.ptr(x), L.ptr(dx), x.numel(), L.dt(x), L.stream()),
            'gelu_bwd')
    return dx


def tanh_fwd(x):
    y = torch.empty_like(x)
    L.check(_lib().passl_hip_tanh_fwd(L.ptr(x), L.ptr(y), x.numel(), L.dt(x), L.stream()), 'tanh_fwd')
    return y


def tanh_bwd(dy, x):
    dx = torch.empty_like(x)
    L.check(_lib().passl_hip_tanh_bwd(L.ptr(dy), L.ptr(x), L.ptr(dx), x.numel(), L.dt(x), L.stream()),
            'tanh_bwd')
    return dx


# the envelope of csrc/attention.hip / attention_bf16.hip (shape_ok): model constructors check it up front
ATTENTION_HEAD_DIMS = frozenset((32, 64))
ATTENTION_MAX_TOKENS = 208


def attention_fwd(qkv, B, T, H, DH, scale, causal=False):
    """qkv [B*T, 3*H*DH] -> out [B*T, H*DH], lse [B, H, T]."""
    out = torch.empty(B * T, H * DH, dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty(B, H, T, dtype=torch.float32, device=qkv.device)
    L.check(_lib().passl_hip_attention_fwd(L.ptr(qkv), L.ptr(out), L.ptr(lse), B, T, H, DH, scale,
                                           int(causal), L.dt(qkv), L.stream()), 'attention_fwd')
    return out, lse


def attention_bwd(qkv, out, dout, lse, B, T, H, DH, scale, causal=False):
    dqkv = torch.empty_like(qkv)
    L.check(_lib().passl_hip_attention_bwd(L.ptr(qkv), L.ptr(out), L.ptr(dout), L.ptr(lse), L.ptr(dqkv),
                                           B, T, H, DH, scale, int(causal), L.dt(qkv), L.stream()),
            'attention_bwd')
    return dqkv


def mae_mask(noise, len_keep):
    B, Ln = noise.shape
    dev = noise.device
    ids_keep = torch.empty(B, len_keep, dtype=torch.int32, device=dev)
    ids_restore = torch.empty(B, Ln, dtype=torch.int32, device=dev)
    mask = torch.empty(B, Ln, dtype=torch.float32, device=dev)
    L.check(_lib().passl_hip_mae_mask(L.ptr(noise), B, Ln, len_keep, L.ptr(ids_keep), L.ptr(ids_restore),
                                      L.ptr(mask), L.stream()), 'mae_mask')
    return ids_keep, ids_restore, mask


def mae_gather(x, cls, pos, ids_keep, B, Ln):
    K, D = ids_keep.shape[1], x.shape[-1]
    out = torch.empty(B * (K + 1), D, dtype=x.dtype, device=x.device)
    L.check(_lib().passl_hip_mae_gather(L.ptr(x), L.ptr(cls), L.ptr(pos), L.ptr(ids_keep), L.ptr(out), B,
                                        Ln, K, D, L.dt(x), L.stream()), 'mae_gather')
    return out


def mae_gather_bwd(dout, ids_restore, dcls, B, Ln, K):
    D = dout.shape[-1]
    dx = torch.empty(B * Ln, D, dtype=dout.dtype, device=dout.device)
    L.check(_lib().passl_hip_mae_gather_bwd(L.ptr(dout), L.ptr(ids_restore), L.ptr(dx), L.ptr(dcls), B, Ln,
                                            K, D, L.dt(dout), L.stream()), 'mae_gather_bwd')
    return dx


def mae_unshuffle(x, mask_token, pos, ids_restore, B, K):
    Ln, D = ids_restore.shape[1], x.shape[-1]
    out = torch.empty(B * (Ln + 1), D, dtype=x.dtype, device=x.device)
    L.check(_lib().passl_hip_mae_unshuffle(L.ptr(x), L.ptr(mask_token), L.ptr(pos), L.ptr(ids_restore),
                                           L.ptr(out), B, Ln, K, D, L.dt(x), L.stream()), 'mae_unshuffle')
    return out


def mae_unshuffle_bwd(dout, ids_keep, ids_restore, dmask_token, B):
    K, Ln, D = ids_keep.shape[1], ids_restore.shape[1], dout.shape[-1]
    dx = torch.empty(B * (K + 1), D, dtype=dout.dtype, device=dout.device)
    ws = workspace.get(1024 * D, dout.device)
    L.check(_lib().passl_hip_mae_unshuffle_bwd(L.ptr(dout), L.ptr(ids_keep), L.ptr(ids_restore), L.ptr(dx),
                                               L.ptr(dmask_token), B, Ln, K, D, L.dt(dout), L.ptr(ws),
                                               ws.numel(), L.stream()), 'mae_unshuffle_bwd')
    return dx


def patchify(img, p, dtype):
    B, Cc, H, W = img.shape
    out = torch.empty(B * (H // p) * (W // p), p * p * Cc, dtype=dtype, device=img.device)
    L.check(_lib().passl_hip_patchify(L.ptr(img), L.ptr(out), B, Cc, H, W, p, L.dt(dtype), L.stream()),
            'patchify')
    return out


def mae_loss_fwd(img, pred, mask, p, norm_pix, denom):
    B, Cc, H, W = img.shape
    loss = torch.empty(1, dtype=torch.float32, device=img.device)
    ws = workspace.get(B * ((H // p) * (W // p) + 1), img.device)
    L.check(_lib().passl_hip_mae_loss_fwd(L.ptr(img), L.ptr(pred), L.ptr(mask), L.ptr(loss), B, Cc, H, W, p,
                                          1 if norm_pix else 0, denom, L.ptr(ws), ws.numel(), L.stream()),
            'mae_loss_fwd')
    return loss


def mae_loss_bwd(img, pred, mask, gscale, p, norm_pix, denom):
    B, Cc, H, W = img.shape
    dpred = torch.empty_like(pred)
    L.check(_lib().passl_hip_mae_loss_bwd(L.ptr(img), L.ptr(pred), L.ptr(mask), L.ptr(gscale), L.ptr(dpred),
                                          B, Cc, H, W, p, 1 if norm_pix else 0, denom, L.stream()),
            'mae_loss_bwd')
    return dpred


def adamw(p, g, m, v, lr, b1, b2, eps, wd, b1pow, b2pow, grad_scale=1.0):
    L.check(_lib().passl_hip_adamw(L.ptr(p), L.ptr(g), L.ptr(m), L.ptr(v), p.numel(), lr, b1, b2, eps, wd,
                                   b1pow, b2pow, grad_scale, L.stream()), 'adamw')


def adamw_dev(p, g, m, v, hyper, b1, b2, eps, wd, grad_scale=1.0):
    """hyper: device float32 [lr, beta1^t, beta2^t], read when the kernel runs."""
    L.check(_lib().passl_hip_adamw_dev(L.ptr(p), L.ptr(g), L.ptr(m), L.ptr(v), p.numel(), L.ptr(hyper), b1, b2, eps,
                                       wd, grad_scale, L.stream()), 'adamw_dev')


# ------------------------------------------------------------------ CLIP
def quick_gelu_fwd(x):
    y = torch.empty_like(x)
    L.check(_lib().passl_hip_quick_gelu_fwd(L.ptr(x), L.ptr(y), x.numel(), L.dt(x), L.stream()),
            'quick_gelu_fwd')
    return y


def quick_gelu_bwd(dy, x):
    dx = torch.empty_like(x)
    L.check(_lib().passl_hip_quick_gelu_bwd(L.ptr(dy), L.ptr(x), L.ptr(dx), x.numel(), L.dt(x), L.stream()),
            'quick_gelu_bwd')
    return dx


def embed_fwd(text, table, pos, dtype):
    """text int64 [B,T]; table fp32 [V,C]; pos fp32 [T,C] -> [B*T, C] in `dtype`."""
    B, T = text.shape
    V, Cc = table.shape
    out = torch.empty(B * T, Cc, dtype=dtype, device=table.device)
    L.check(_lib().passl_hip_embed_fwd(L.ptr(text), L.ptr(table), L.ptr(pos), L.ptr(out), B, T, Cc, V,
                                       L.dt(out), L.stream()), 'embed_fwd')
    return out


_embed_acc = {}


def embed_bwd(text, dout, dtable, dpos):
    """dtable[text] += dout (exact fixed-point scatter-add), dpos += sum_b dout: see include/passl_hip.h."""
    B, T = text.shape
    V, Cc = dtable.shape
    lib = _lib()
    # the persistent accumulator of the fixed-point scatter: zero once, every call leaves it zero
    key = (dout.device.type, dout.device.index, V, Cc)
    acc = _embed_acc.get(key)
    if acc is None:
        acc = torch.zeros(int(lib.passl_hip_embed_bwd_acc_bytes(V, Cc)) // 8, dtype=torch.int64, device=dout.device)
        _embed_acc[key] = acc
    ws = workspace.get(int(lib.passl_hip_embed_bwd_ws_floats(B, T, Cc)), dout.device)
    L.check(lib.passl_hip_embed_bwd(L.ptr(text), L.ptr(dout), L.ptr(dtable), L.ptr(dpos), B, T, Cc, V,
                                    L.dt(dout), L.ptr(acc), acc.numel() * 8, L.ptr(ws), ws.numel(), L.stream()),
            'embed_bwd')


def gather_rows(x, idx):
    n, Cc = idx.numel(), x.shape[-1]
    out = torch.empty(n, Cc, dtype=x.dtype, device=x.device)
    L.check(_lib().passl_hip_gather_rows(L.ptr(x), L.ptr(idx), L.ptr(out), n, Cc, L.dt(x), L.stream()),
            'gather_rows')
    return out


def scatter_rows(dout, idx, rows_total):
    n, Cc = dout.shape
    dx = torch.empty(rows_total, Cc, dtype=dout.dtype, device=dout.device)
    L.check(_lib().passl_hip_scatter_rows(L.ptr(dout), L.ptr(idx), L.ptr(dx), n, rows_total, Cc, L.dt(dout),
                                          L.stream()), 'scatter_rows')
    return dx


def eot_index(text):
    B, T = text.shape
    idx = torch.empty(B, dtype=torch.int32, device=text.device)
    L.check(_lib().passl_hip_eot_index(L.ptr(text), B, T, L.ptr(idx), L.stream()), 'eot_index')
    return idx


def clip_logits_fwd(img, txt, logit_scale, clip_lo=-4.6, clip_hi=4.6):
    """-> logits [B,B] (= image_logits; text_logits is its transpose), ws (workspace for the backward).
    logit_scale (the fp32 parameter) is clipped in place after use."""
    B, Dd = img.shape
    ws = torch.empty(int(_lib().passl_hip_clip_logits_ws_floats(B, Dd)), dtype=torch.float32, device=img.device)
    logits = torch.empty(B, B, dtype=torch.float32, device=img.device)
    L.check(_lib().passl_hip_clip_logits_fwd(L.ptr(img), L.ptr(txt), L.ptr(logit_scale), B, Dd, clip_lo,
                                             clip_hi, L.ptr(ws), L.ptr(logits), L.stream()), 'clip_logits_fwd')
    return logits, ws


def clip_logits_bwd(dlogits, logits, ws, Dd, dlogit_scale):
    B = logits.shape[0]
    dimg = torch.empty(B, Dd, dtype=torch.float32, device=ws.device)
    dtxt = torch.empty(B, Dd, dtype=torch.float32, device=ws.device)
    scratch = workspace.get(256, ws.device)
    L.check(_lib().passl_hip_clip_logits_bwd(L.ptr(dlogits), L.ptr(logits), L.ptr(ws), B, Dd, L.ptr(dimg),
                                             L.ptr(dtxt), L.ptr(dlogit_scale), L.ptr(scratch), L.stream()),
            'clip_logits_bwd')
    return dimg, dtxt


def clip_scale(logit_scale, clip_lo=-4.6, clip_hi=4.6):
    """-> alpha (device scalar) = exp(logit_scale) as used by this step; logit_scale is clipped in place."""
    alpha = torch.empty(1, dtype=torch.float32, device=logit_scale.device)
    L.check(_lib().passl_hip_clip_scale(L.ptr(logit_scale), L.ptr(alpha), clip_lo, clip_hi, L.stream()), 'clip_scale')
    return alpha


def gemm_f32_nt(a, b, alpha):
    """alpha * a [M,K] @ b[N,K]^T -> [M,N] fp32 (exact-fp32 MFMA)."""
    M, K = a.shape
    N = b.shape[0]
    c = torch.empty(M, N, dtype=torch.float32, device=a.device)
    L.check(_lib().passl_hip_gemm_f32_nt(L.ptr(a), L.ptr(b), L.ptr(c), M, N, K, L.ptr(alpha), L.stream()), 'gemm_f32_nt')
    return c


def gemm_f32_gx(g, x, alpha, trans=False):
    """alpha * op(g) @ x[K,N]: op(g) = g [M,K] or (trans) g^T with g [K,M]."""
    K, N = x.shape
    M = g.shape[1] if trans else g.shape[0]
    c = torch.empty(M, N, dtype=torch.float32, device=x.device)
    L.check(_lib().passl_hip_gemm_f32_gx(L.ptr(g), L.ptr(x), L.ptr(c), M, N, K, 1 if trans else 0, L.ptr(alpha),
                                         L.stream()), 'gemm_f32_gx')
    return c


def dot_acc(a, b, out):
    """out[0] += sum(a * b), fixed summation order."""
    ws = workspace.get(256, a.device)
    L.check(_lib().passl_hip_dot_acc(L.ptr(a), L.ptr(b), a.numel(), L.ptr(out), L.ptr(ws), L.stream()), 'dot_acc')


def clip_ce_fwd(logits):
    """-> out[3] = (img_loss, text_loss, loss), lse [2B]."""
    B = logits.shape[0]
    lse = torch.empty(2 * B, dtype=torch.float32, device=logits.device)
    out = torch.empty(3, dtype=torch.float32, device=logits.device)
    ws = workspace.get(2 * B, logits.device)
    L.check(_lib().passl_hip_clip_ce_fwd(L.ptr(logits), B, L.ptr(lse), L.ptr(out), L.ptr(ws), ws.numel(),
                                         L.stream()), 'clip_ce_fwd')
    return out, lse


def clip_ce_bwd(logits, lse, gloss):
    dlogits = torch.empty_like(logits)
    L.check(_lib().passl_hip_clip_ce_bwd(L.ptr(logits), L.ptr(lse), L.ptr(gloss), logits.shape[0],
                                         L.ptr(dlogits), L.stream()), 'clip_ce_bwd')
    return dlogits


# ------------------------------------------------------------------ linear probe
def softmax_ce_fwd(scores, labels):
    """scores fp32 [N,C], labels int64 [N] -> out[3] = (loss, acc1 %, acc5 %), lse [N]."""
    N, Cc = scores.shape
    lse = torch.empty(N, dtype=torch.float32, device=scores.device)
    out = torch.empty(3, dtype=torch.float32, device=scores.device)
    ws = workspace.get(3 * N, scores.device)
    L.check(_lib().passl_hip_softmax_ce_fwd(L.ptr(scores), L.ptr(labels), N, Cc, L.ptr(lse), L.ptr(out),
                                            L.ptr(ws), ws.numel(), L.stream()), 'softmax_ce_fwd')
    return out, lse


def softmax_ce_bwd(scores, lse, labels, gloss):
    N, Cc = scores.shape
    ds = torch.empty_like(scores)
    L.check(_lib().passl_hip_softmax_ce_bwd(L.ptr(scores), L.ptr(lse), L.ptr(labels), L.ptr(gloss), N, Cc,
                                            L.ptr(ds), L.stream()), 'softmax_ce_bwd')
    return ds
