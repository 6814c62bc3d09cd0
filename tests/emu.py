"""CPU emulator of the addressing rules of passl_hip_conv_igemm / passl_hip_conv_wgrad /
passl_hip_pack_weights (include/passl_hip.h), used to validate host-side plans without a GPU.
TEST INFRASTRUCTURE ONLY — it is never used to produce product results."""
import torch


def emu_pack(pack, w_krsc):
    """w_krsc: [K,R,S,C] -> packed tensor as the pack job defines it."""
    K, R, S, C = w_krsc.shape
    inner_src = K if pack.transpose else C
    inner = pack.c_pad if pack.c_pad > 0 else inner_src
    outer = C if pack.transpose else K
    dst = torch.zeros(outer, pack.TR, pack.TS, inner, dtype=w_krsc.dtype)
    for tr in range(pack.TR):
        r = pack.r_base + tr * pack.r_step
        if not (0 <= r < R):
            continue
        for ts in range(pack.TS):
            s = pack.s_base + ts * pack.s_step
            if not (0 <= s < S):
                continue
            blk = w_krsc[:, r, s, :]                 # [K, C]
            if pack.transpose:
                dst[:, tr, ts, :inner_src] = blk.t()
            else:
                dst[:, tr, ts, :inner_src] = blk
    return dst


def _gather(d, a_flat, r, s):
    n = torch.arange(d.N).view(-1, 1, 1)
    op = torch.arange(d.OP).view(1, -1, 1)
    oq = torch.arange(d.OQ).view(1, 1, -1)
    ih = op * d.sh + r - d.ph
    iw = oq * d.sw + s - d.pw
    valid = ((ih >= 0) & (ih < d.IH) & (iw >= 0) & (iw < d.IW)).expand(d.N, d.OP, d.OQ)
    addr = n * d.a_sn + ih.clamp(0, d.IH - 1) * d.a_sh + iw.clamp(0, d.IW - 1) * d.a_sw
    addr = addr.expand(d.N, d.OP, d.OQ)
    idx = addr.reshape(-1, 1) + torch.arange(d.C).view(1, -1)
    g = a_flat[idx]                                   # [M, C]
    return g * valid.reshape(-1, 1).to(g.dtype)


def emu_conv(d, a_flat, bmat, y_flat, scale=None, shift=None, res_flat=None, relu=False):
    """Writes the conv result into y_flat (1-D) exactly where the kernel would."""
    M = d.N * d.OP * d.OQ
    acc = torch.zeros(M, d.NCOLS, dtype=a_flat.dtype)
    B = bmat.reshape(d.NCOLS, d.R * d.S * d.C)
    for r in range(d.R):
        for s in range(d.S):
            g = _gather(d, a_flat, r, s)
            k0 = (r * d.S + s) * d.C
            acc += g @ B[:, k0:k0 + d.C].t()
    if scale is not None:
        acc = acc * scale.view(1, -1)
    if shift is not None:
        acc = acc + shift.view(1, -1)
    n = torch.arange(d.N).view(-1, 1, 1)
    op = torch.arange(d.OP).view(1, -1, 1)
    oq = torch.arange(d.OQ).view(1, 1, -1)
    yo = (d.y_off + n * d.y_sn + op * d.y_sh + oq * d.y_sw).expand(d.N, d.OP, d.OQ).reshape(-1, 1)
    idx = yo + torch.arange(d.NCOLS).view(1, -1)
    if res_flat is not None:
        acc = acc + res_flat[idx]
    if relu:
        acc = acc.clamp_min(0)
    y_flat[idx] = acc
    return y_flat


def emu_wgrad(d, a_flat, dy):
    """dy: [M, NCOLS]  ->  dw [NCOLS, R*S*C]"""
    dw = torch.zeros(d.NCOLS, d.R * d.S * d.C, dtype=a_flat.dtype)
    for r in range(d.R):
        for s in range(d.S):
            g = _gather(d, a_flat, r, s)
            k0 = (r * d.S + s) * d.C
            dw[:, k0:k0 + d.C] = dy.t() @ g
    return dw
