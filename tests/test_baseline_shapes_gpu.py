"""Head / attention / masking kernels at the shapes BASELINE.json configs[2..4] are quoted on (round-5 verdict,
"what's missing" #4: these kernels were only checked at toy sizes).  Everything here is head- or block-level, so the
fp64 oracle on the host is affordable; whole-model steps at these batch sizes stay with the golden fixtures.

  * configs[2] SimCLR, global batch 4096 on 8 ranks: this rank's 512 rows against the 4096 gathered columns of both
    views (reference passl_v110/modeling/heads/simclr_contrastive_head.py:42-102 with the CO2 term);
  * configs[4] CLIP, global batch 8192 on 8 ranks: 1024 local image / text features against 8192 gathered ones,
    row cross-entropies with labels arange(B) + B*rank (reference clip.py:320-338, clip_head.py:27-35, gather pattern
    of passl/models/mocov3.py:187-198);
  * configs[3] MAE ViT-B/16 at 256 images / GPU: masking ranks, masked-patch loss fwd / bwd (passl/models/mae.py:126-150,
    268-284), attention fwd / bwd for the encoder (50 tokens, 12 heads x 64) and the decoder (197 tokens, 16 x 32) at
    B = 256 (passl/models/vision_transformer.py:142-156) — also CLIP ViT-B/16's image tower shape (197, 12 x 64).

Tolerances are those of the small-shape tests of the same kernels (tests/test_simclr_gpu.py, test_clip_gpu.py,
test_mae_gpu.py)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import mae as M                    # noqa: E402
from passl_amd.hip import ops                  # noqa: E402

DEV = 'cuda'


def relmax(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


def test_ntxent_co2_rank_rows_512_against_4096_gathered_columns():
    from test_simclr_gpu import _head_general, _unit
    gen = torch.Generator().manual_seed(4096)
    B, BL, rank, T = 512, 4096, 3, 0.1
    roff = rank * B
    a_all, b_all = _unit(BL, gen).double(), _unit(BL, gen).double()
    # the two views of one image are correlated (a trained encoder's regime: the positive logit stands out)
    b_all = F.normalize(b_all + 0.8 * a_all, dim=1)
    h1 = a_all[roff:roff + B].clone().requires_grad_(True)
    h2 = b_all[roff:roff + B].clone().requires_grad_(True)
    A = a_all.clone().requires_grad_(True)
    Bm = b_all.clone().requires_grad_(True)
    loss, acc = _head_general(h1, h2, A, Bm, roff, T)
    (loss * 0.5).backward()
    f = lambda t: t.detach().float().to(DEV).contiguous()
    out, rs = ops.ntxent_fwd(f(h1), f(h2), f(A), f(Bm), roff, T, 3.0)
    assert abs(float(out[0]) - float(loss)) < 2e-4 * max(1.0, abs(float(loss)))      # fp32 path: within 1e-3 (north_star)
    assert abs(float(out[1]) - float(acc)) < 1e-6
    da, db, dA, dB = ops.ntxent_bwd(f(h1), f(h2), f(A), f(Bm), rs, torch.tensor([0.5], device=DEV), roff, T, 3.0)
    for got, ref in ((da, h1.grad), (db, h2.grad), (dA, A.grad), (dB, Bm.grad)):
        assert got.shape == ref.shape
        s = max(float(ref.abs().max()), 1e-12)
        assert float((got.cpu().double() - ref).abs().max()) / s < 3e-4
    # second run: bit-identical (fixed-order reductions)
    out2, _ = ops.ntxent_fwd(f(h1), f(h2), f(A), f(Bm), roff, T, 3.0)
    assert torch.equal(out, out2)


def test_clip_cross_rank_infonce_1024_local_against_8192_gathered():
    gen = torch.Generator().manual_seed(8192)
    B, W, D, rank = 1024, 8, 512, 5
    img_all = (torch.randn(W * B, D, generator=gen) * 3)
    txt_all = (torch.randn(W * B, D, generator=gen) * 0.5 + 0.3 * img_all / 3)
    lo = rank * B
    fi = img_all[lo:lo + B].clone().requires_grad_(True)
    ft = txt_all[lo:lo + B].clone().requires_grad_(True)
    s = torch.tensor([math.log(1 / 0.07)], dtype=torch.float64, requires_grad=True)
    n = lambda x: x.double() / x.double().norm(dim=-1, keepdim=True)
    Ia = n(img_all).clone().requires_grad_(True)     # the gathered (already normalised) copies: leaves of their own
    Ta = n(txt_all).clone().requires_grad_(True)
    li = s.exp() * n(fi) @ Ta.t()
    lt = s.exp() * n(ft) @ Ia.t()
    lab = torch.arange(B) + lo
    loss_i, loss_t = F.cross_entropy(li, lab), F.cross_entropy(lt, lab)
    (loss_i + loss_t).backward()

    d = lambda t: t.detach().float().to(DEV).contiguous()
    sd = torch.tensor([math.log(1 / 0.07)], device=DEV)
    img_n, img_norm = ops.l2norm_fwd(d(fi), 0.0)
    txt_n, txt_norm = ops.l2norm_fwd(d(ft), 0.0)
    alpha = ops.clip_scale(sd)
    Iad, Tad = d(Ia), d(Ta)
    gli = ops.gemm_f32_nt(img_n, Tad, alpha)
    glt = ops.gemm_f32_nt(txt_n, Iad, alpha)
    assert gli.shape == (B, W * B)
    assert relmax(gli, li.detach()) < 3e-6 and relmax(glt, lt.detach()) < 3e-6
    labd = lab.to(DEV)
    oi, lse_i = ops.softmax_ce_fwd(gli, labd)
    ot, lse_t = ops.softmax_ce_fwd(glt, labd)
    assert abs(float(oi[0]) - float(loss_i)) < 2e-5 * max(1.0, float(loss_i))
    assert abs(float(ot[0]) - float(loss_t)) < 2e-5 * max(1.0, float(loss_t))
    one = torch.ones(1, device=DEV)
    dli = ops.softmax_ce_bwd(gli, lse_i, labd, one)
    dlt = ops.softmax_ce_bwd(glt, lse_t, labd, one)
    # local rows' role, through the normalisation
    dimg = ops.l2norm_bwd(ops.gemm_f32_gx(dli, Tad, alpha), img_n, img_norm, torch.float32)
    dtxt = ops.l2norm_bwd(ops.gemm_f32_gx(dlt, Iad, alpha), txt_n, txt_norm, torch.float32)
    assert relmax(dimg, fi.grad) < 1e-4 and relmax(dtxt, ft.grad) < 1e-4
    # column role: what this rank contributes to the reduce-scatter over the gathered copies
    dTa = ops.gemm_f32_gx(dli, img_n, alpha, trans=True)
    dIa = ops.gemm_f32_gx(dlt, txt_n, alpha, trans=True)
    assert dTa.shape == (W * B, D)
    assert relmax(dTa, Ta.grad) < 1e-4 and relmax(dIa, Ia.grad) < 1e-4
    # logit_scale gradient: sum dL o L over both matrices
    ds = torch.zeros(1, device=DEV)
    ops.dot_acc(dli, gli, ds)
    ops.dot_acc(dlt, glt, ds)
    assert abs(float(ds) - float(s.grad)) < 1e-4 * max(1.0, abs(float(s.grad)))


def test_mae_masking_and_masked_patch_loss_at_256_images():
    gen = torch.Generator().manual_seed(256)
    B, p, HW = 256, 16, 224
    L = (HW // p) ** 2
    # distinct values per row (fp32 uniforms collide in ~25 % of 256 x 196 draws; ties are covered by test_mae_gpu.py)
    noise = (torch.stack([torch.randperm(L, generator=gen) for _ in range(B)]).float() + 0.5) / L
    keep, mask, restore = M.random_masking_ids(noise, 0.75)
    K = keep.shape[1]
    assert K == 49
    ik, ir, mk = ops.mae_mask(noise.to(DEV), K)
    assert torch.equal(ik.cpu().long(), keep) and torch.equal(ir.cpu().long(), restore) and torch.equal(mk.cpu(), mask)
    img = torch.randn(B, 3, HW, HW, generator=gen)
    tgt = M.patchify(img, p).double()
    tgt = (tgt - tgt.mean(-1, keepdim=True)) / (tgt.var(-1, keepdim=True) + 1e-6) ** .5       # norm_pix_loss
    pred = torch.randn(B, L + 1, p * p * 3, generator=gen).double().requires_grad_(True)
    loss = ((((pred[:, 1:] - tgt) ** 2).mean(-1)) * mask).sum() / mask.sum()
    (loss * 0.9).backward()
    pd = pred.detach().float().to(DEV).reshape(-1, p * p * 3)
    imgd, maskd = img.to(DEV), mask.to(DEV)
    got = ops.mae_loss_fwd(imgd, pd, maskd, p, True, float(mask.sum()))
    assert abs(float(got) - float(loss)) < 2e-5 * max(1.0, float(loss))
    dp = ops.mae_loss_bwd(imgd, pd, maskd, torch.tensor([0.9], device=DEV), p, True, float(mask.sum()))
    assert relmax(dp.reshape(B, L + 1, -1), pred.grad) < 1e-4
    assert float(dp.reshape(B, L + 1, -1)[:, 0].abs().max()) == 0
    assert float(ops.mae_loss_fwd(imgd, pd, maskd, p, True, float(mask.sum()))) == float(got)    # reproducible


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('T,H,DH', [(50, 12, 64), (197, 16, 32), (197, 12, 64)])
def test_attention_fwd_bwd_at_batch_256(dtype, T, H, DH):
    B, CH = 256, 32
    gen = torch.Generator().manual_seed(T * H)
    qkv = torch.randn(B, T, 3, H, DH, generator=gen).to(dtype).float()
    dout = torch.randn(B * T, H * DH, generator=gen).to(dtype).float()
    scale = DH ** -0.5
    qd = qkv.reshape(B * T, 3 * H * DH).to(DEV).to(dtype)
    od, lse = ops.attention_fwd(qd, B, T, H, DH, scale)
    dq = ops.attention_bwd(qd, od, dout.to(DEV).to(dtype), lse, B, T, H, DH, scale)
    od, lse, dq = od.float().cpu().reshape(B, T, H * DH), lse.cpu(), dq.float().cpu().reshape(B, T, 3, H, DH)
    t_out, t_dq = (2e-5, 1e-4) if dtype == torch.float32 else (2e-2, 3e-2)
    for b0 in range(0, B, CH):                  # the fp64 reference one chunk of images at a time (attention is per image)
        x = qkv[b0:b0 + CH].double().requires_grad_(True)
        q, k, v = [x[:, :, i].permute(0, 2, 1, 3) for i in range(3)]
        sc = q @ k.transpose(-1, -2) * scale
        out = (torch.softmax(sc, dim=-1) @ v).permute(0, 2, 1, 3).reshape(CH, T, H * DH)
        out.backward(dout.reshape(B, T, H * DH)[b0:b0 + CH].double())
        assert relmax(od[b0:b0 + CH], out.detach()) < t_out
        assert float((lse.reshape(B, H, T)[b0:b0 + CH].double() - torch.logsumexp(sc, -1).detach()).abs().max()) < 1e-4
        assert relmax(dq[b0:b0 + CH], x.grad) < t_dq
