"""Host-side planning for the implicit-GEMM kernels: turns a convolution's geometry into the
descriptor fields of include/passl_hip.h (forward, data-gradient, weight-gradient) and into
weight-packing jobs.  Pure Python / no GPU: every plan is validated on CPU in
tests/test_plan.py by executing the descriptors with a reference emulator of the kernel's
addressing rules and comparing with torch's conv2d autograd.

Reference call sites being planned: nn.Conv2D uses at
passl_v110/modeling/backbones/resnetimagenet.py:114-131 (bottleneck), :190-195 (stem),
:216-224 (downsample); nn.Linear at passl_v110/modeling/necks/base_neck.py:80-85.
"""
import os
from dataclasses import dataclass
from typing import List, Optional


@dataclass(frozen=True)
class ConvGeom:
    cin: int
    cout: int
    k: int          # square kernel
    stride: int
    pad: int

    def out_hw(self, h, w):
        return ((h + 2 * self.pad - self.k) // self.stride + 1,
                (w + 2 * self.pad - self.k) // self.stride + 1)


@dataclass
class Pack:
    """dst[outer][TR][TS][inner] = src[k][r_base+tr*r_step][s_base+ts*s_step][c]
    (transpose: outer=c, inner=k; else outer=k, inner=c).  `size` in elements."""
    TR: int
    TS: int
    r_base: int
    r_step: int
    s_base: int
    s_step: int
    transpose: int
    c_pad: int = 0
    size: int = 0
    dst_off: int = -1          # filled by the packer


@dataclass
class Desc:
    """Fields of passl_conv_desc that do not depend on pointers."""
    N: int
    OP: int
    OQ: int
    NCOLS: int
    R: int
    S: int
    C: int
    IH: int
    IW: int
    sh: int
    sw: int
    ph: int
    pw: int
    a_sn: int
    a_sh: int
    a_sw: int
    y_sn: int
    y_sh: int
    y_sw: int
    y_off: int = 0             # element offset added to the output base pointer
    pack: Optional[Pack] = None


def dense_strides(h, w, c):
    return h * w * c, w * c, c


def fwd_desc(g: ConvGeom, N, H, W) -> Desc:
    """y[n,p,q,k] = sum x[n, p*st + r - pad, q*st + s - pad, c] w[k,r,s,c]."""
    P, Q = g.out_hw(H, W)
    a = dense_strides(H, W, g.cin)
    y = dense_strides(P, Q, g.cout)
    return Desc(N=N, OP=P, OQ=Q, NCOLS=g.cout, R=g.k, S=g.k, C=g.cin, IH=H, IW=W,
                sh=g.stride, sw=g.stride, ph=g.pad, pw=g.pad,
                a_sn=a[0], a_sh=a[1], a_sw=a[2], y_sn=y[0], y_sh=y[1], y_sw=y[2],
                pack=Pack(TR=g.k, TS=g.k, r_base=0, r_step=1, s_base=0, s_step=1, transpose=0,
                          size=g.cout * g.k * g.k * g.cin))


def _taps(a, pad, k, st):
    """For input rows h = st*i + a:  contributing filter taps and the dy row offset.
    Returns (J, r_base, r_step, off) such that tap t in [0,J) uses filter row
    r = r_base + t*r_step and reads dy row p = i + t + off.  J == 0 -> no contribution."""
    r_min = (a + pad) % st
    if r_min >= k:
        return 0, 0, 0, 0
    J = (k - r_min + st - 1) // st
    off = (a + pad - r_min) // st - (J - 1)
    return J, r_min + st * (J - 1), -st, off


def dgrad_plan(g: ConvGeom, N, H, W, packs=None):
    """dx[n,h,w,c] = sum_{k,r,s} dy[n,(h+pad-r)/st,(w+pad-s)/st,k] w[k,r,s,c] as one
    forward-style conv over dy per residue class (a,b) = (h % st, w % st), each writing the
    sub-lattice dx[:, a::st, b::st, :].  stride 1 -> a single class = the classic flipped-filter
    conv.  Returns (descs, needs_zero_fill): classes with no contributing tap are skipped and
    dx must then be zero-filled first.  `packs` (dict (a,b) -> Pack) lets a caller reuse Pack
    objects already registered with a WeightPacker (pack geometry is size independent)."""
    P, Q = g.out_hw(H, W)
    st = g.stride
    dy = dense_strides(P, Q, g.cout)
    dx = dense_strides(H, W, g.cin)
    descs: List[Desc] = []
    skipped = False
    for a in range(st):
        Ja, rb, rstep, offh = _taps(a, g.pad, g.k, st)
        nh = (H - a + st - 1) // st          # rows h = a, a+st, ... < H
        for b in range(st):
            Jb, sb, sstep, offw = _taps(b, g.pad, g.k, st)
            nw = (W - b + st - 1) // st
            if nh <= 0 or nw <= 0:
                continue
            if Ja == 0 or Jb == 0:
                skipped = True
                continue
            descs.append(Desc(
                N=N, OP=nh, OQ=nw, NCOLS=g.cin, R=Ja, S=Jb, C=g.cout, IH=P, IW=Q,
                sh=1, sw=1, ph=-offh, pw=-offw,
                a_sn=dy[0], a_sh=dy[1], a_sw=dy[2],
                y_sn=dx[0], y_sh=dx[1] * st, y_sw=dx[2] * st,
                y_off=a * dx[1] + b * dx[2],
                pack=(packs[(a, b)] if packs is not None else
                      Pack(TR=Ja, TS=Jb, r_base=rb, r_step=rstep, s_base=sb, s_step=sstep,
                           transpose=1, size=g.cin * Ja * Jb * g.cout))))
    return descs, skipped


def dgrad_packs(g: ConvGeom):
    """Size-independent Pack per residue class (a,b) that has contributing taps."""
    out = {}
    for a in range(g.stride):
        Ja, rb, rstep, _ = _taps(a, g.pad, g.k, g.stride)
        for b in range(g.stride):
            Jb, sb, sstep, _ = _taps(b, g.pad, g.k, g.stride)
            if Ja and Jb:
                out[(a, b)] = Pack(TR=Ja, TS=Jb, r_base=rb, r_step=rstep, s_base=sb,
                                   s_step=sstep, transpose=1, size=g.cin * Ja * Jb * g.cout)
    return out


def wgrad_desc(g: ConvGeom, N, H, W) -> Desc:
    """dw[k,r,s,c] = sum_{n,p,q} dy[n,p,q,k] x[n, p*st+r-pad, q*st+s-pad, c]  (same gather as fwd)."""
    d = fwd_desc(g, N, H, W)
    d.pack = None
    return d


# ---- stem (7x7 stride-2 pad-3, Cin=3): read from a zero-padded NHWC image with 4 channels whose
# row is viewed as (pixel-pair, 8 channels); one filter row = 32 contiguous elements
# (8 taps x 4 channels, tap 7 and channel 3 carry zero weights).
STEM_K, STEM_PAD, STEM_CP, STEM_ROW = 7, 3, 4, 32


def stem_padded_hw(H, W):
    Hp = H + 2 * STEM_PAD
    Wp = W + 2 * STEM_PAD
    Wp += Wp & 1
    # the widest read is 8 pixels starting at column 2*(Q-1)
    Q = (W + 2 * STEM_PAD - STEM_K) // 2 + 1
    need = 2 * (Q - 1) + 8
    if Wp < need:
        Wp = need + (need & 1)
    return Hp, Wp


def stem_desc(cout, N, H, W) -> Desc:
    Hp, Wp = stem_padded_hw(H, W)
    P = (H + 2 * STEM_PAD - STEM_K) // 2 + 1
    Q = (W + 2 * STEM_PAD - STEM_K) // 2 + 1
    y = dense_strides(P, Q, cout)
    return Desc(N=N, OP=P, OQ=Q, NCOLS=cout, R=STEM_K, S=1, C=STEM_ROW, IH=Hp, IW=Q,
                sh=2, sw=1, ph=0, pw=0,
                a_sn=Hp * Wp * STEM_CP, a_sh=Wp * STEM_CP, a_sw=2 * STEM_CP,
                y_sn=y[0], y_sh=y[1], y_sw=y[2],
                pack=Pack(TR=STEM_K, TS=8, r_base=0, r_step=1, s_base=0, s_step=1, transpose=0,
                          c_pad=STEM_CP, size=cout * STEM_K * 8 * STEM_CP))


# Grid targets of the weight-gradient launches (workgroups per launch).  Inside a training step these kernels run on the
# side stream NEXT TO the main chain's kernels; every reduction slice writes a full fp32 tile of dW to a slab that
# slab_reduce reads again (with 512 workgroups per launch: 3.4 GB of the R50 step's 100 GB).  Round 6, same-box A/Bs inside
# the steps (profiles/r06_wgrad_grid_ab.txt):
#   * convolutional networks — the launch shares the memory system with bandwidth-bound BatchNorm / 1x1 kernels: ONE
#     workgroup per CU instead of two takes the MoCo step from 23.13 / 22.96 / 23.06 to 22.82 / 22.63 / 22.85 ms on three boxes
#     (spatially tiled 3x3 kernel -0.2 ms, LDS-DMA kernel -0.1), SimCLR 35.37 -> 35.16; 192 is level, 128 too few
#     (+0.2 ... +1.2 ms); the last stage's layers, although MFMA-bound by their shapes, prefer it too;
#   * transformer Linears (M = tokens, no spatial extent) — the launch runs beside MFMA-bound GEMMs and needs two workgroups
#     per CU: with 256 MAE loses 9 %, CLIP 10 %.
# So: layers with a spatial extent 256, Linears 512.  PASSL_WGRAD_TARGET_BLOCKS / PASSL_WGRAD_HALO_TARGET_BLOCKS override
# (experiments).
_WGRAD_HALO_TARGET = int(os.environ.get('PASSL_WGRAD_HALO_TARGET_BLOCKS', '0') or 0)
_WGRAD_TARGET = int(os.environ.get('PASSL_WGRAD_TARGET_BLOCKS', '0') or 0)


def wgrad_halo_splits(N, IH, IW, ncols, C, target_blocks=256):
    """Slices for the spatially tiled 3x3 weight-gradient kernel (csrc/conv_wgrad_halo.inc): its grid is one
    workgroup per 64 x 64 block of dW and slice (all nine taps), a k-tile is one 8 x 8 patch.  Stand-alone at 64 -> 64 @56
    (profiles/r04_kbench_halo_experiment.txt): 256 slices 83 us, 512 slices 77 us, 1024 slices 104 us; inside the step
    256 wins (see above)."""
    blocks = ((ncols + 63) // 64) * ((C + 63) // 64)
    patches = N * ((IH + 7) // 8) * ((IW + 7) // 8)
    if _WGRAD_HALO_TARGET:
        target_blocks = _WGRAD_HALO_TARGET
    return max(1, min(patches, target_blocks // max(blocks, 1)))


def wgrad_target_blocks(spatial):
    """Workgroups per weight-gradient launch of the LDS-DMA kernel (see the note above): ``spatial`` = the layer is a
    convolution over an image (IH * IW > 1), not a Linear over rows."""
    if _WGRAD_TARGET:
        return _WGRAD_TARGET
    return 256 if spatial else 512


def wgrad_splits(M, ncols, kdim, bkm, target_blocks=512, row_bytes=0):
    """Number of reduction slices so that the grid has ~target_blocks workgroups (``wgrad_target_blocks``: one or two
    per CU).  Every slice adds a full fp32 tile of slab traffic; too few leave the DMA pipeline latency-bound."""
    tiles = ((ncols + 127) // 128) * ((kdim + 127) // 128)
    nk = (M + bkm - 1) // bkm
    s = max(1, min(nk, target_blocks // max(tiles, 1)))
    # the LDS-DMA kernel addresses each M-slice through a rebased 32-bit buffer window: keep a slice's
    # rows below 1 GB of the wider operand (row_bytes = max row pitch of x / dy in bytes)
    if row_bytes:
        s = max(s, min(nk, -(-(M * row_bytes) // (1 << 30))))
    return s
