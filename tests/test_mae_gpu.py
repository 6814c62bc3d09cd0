"""MAE path on a real MI355X through the registries and the C ABI: per-kernel parity against plain
PyTorch fp32/fp64 references (LayerNorm, GELU, fused attention, masking ranks, token gather /
unshuffle, patchify, masked-patch loss, AdamW) and whole training steps against the golden vectors
produced by the reference's own MAE sources (tests/golden/mae_*.npz)."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import mae_util as U                           # noqa: E402
from oracle import mae as M                    # noqa: E402
from passl_amd.hip import ops                  # noqa: E402

DEV = 'cuda'
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
DTYPES = [torch.float32, torch.bfloat16]


def relmax(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


def rnd(t, dtype):
    return t.to(dtype).float()


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('M_,C', [(100, 768), (37, 512), (5, 64), (300, 2048)])
def test_layernorm_fwd_bwd(dtype, M_, C):
    gen = torch.Generator().manual_seed(C)
    x = rnd(torch.randn(M_, C, generator=gen) * 2 + 0.3, dtype).requires_grad_(True)
    g = (torch.rand(C, generator=gen) + 0.5).requires_grad_(True)
    b = torch.randn(C, generator=gen).requires_grad_(True)
    y = F.layer_norm(x.double(), (C,), g.double(), b.double(), 1e-6)
    dy = rnd(torch.randn(M_, C, generator=gen), dtype)
    y.backward(dy.double())
    yd, mean, rstd = ops.layernorm_fwd(x.detach().to(DEV).to(dtype), g.detach().to(DEV), b.detach().to(DEV), 1e-6)
    t = 1e-5 if dtype == torch.float32 else 2e-2
    assert relmax(yd.float(), y.detach()) < t
    dg, db = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    dx = ops.layernorm_bwd(dy.to(DEV).to(dtype), x.detach().to(DEV).to(dtype), g.detach().to(DEV), mean, rstd, dg, db)
    assert relmax(dx.float(), x.grad) < (1e-4 if dtype == torch.float32 else 3e-2)
    assert relmax(dg, g.grad) < 1e-4 and relmax(db, b.grad) < 1e-4
    # residual-fork variant: the gradient of the skip branch is added inside the kernel
    dres = rnd(torch.randn(M_, C, generator=gen), dtype)
    dg2, db2 = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    dx2 = ops.layernorm_bwd(dy.to(DEV).to(dtype), x.detach().to(DEV).to(dtype), g.detach().to(DEV), mean, rstd,
                            dg2, db2, dres=dres.to(DEV).to(dtype))
    assert relmax(dx2.float(), x.grad + dres) < (1e-4 if dtype == torch.float32 else 3e-2)
    assert torch.allclose(dg2, dg, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize('dtype', DTYPES)
def test_gelu(dtype):
    gen = torch.Generator().manual_seed(1)
    x = rnd(torch.randn(64, 3072, generator=gen) * 2, dtype).requires_grad_(True)
    y = F.gelu(x.double())
    dy = rnd(torch.randn(64, 3072, generator=gen), dtype)
    y.backward(dy.double())
    t = 1e-5 if dtype == torch.float32 else 1e-2
    assert relmax(ops.gelu_fwd(x.detach().to(DEV).to(dtype)).float(), y.detach()) < t
    assert relmax(ops.gelu_bwd(dy.to(DEV).to(dtype), x.detach().to(DEV).to(dtype)).float(), x.grad) < t


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('B,T,H,DH', [(3, 50, 12, 64), (2, 197, 16, 32), (4, 5, 4, 32), (2, 17, 2, 32),
                                      (1, 197, 3, 64), (1, 208, 2, 64)])
def test_attention_fwd_bwd(dtype, B, T, H, DH):
    gen = torch.Generator().manual_seed(T)
    qkv = rnd(torch.randn(B, T, 3, H, DH, generator=gen), dtype).requires_grad_(True)
    q, k, v = [qkv.double()[:, :, i].permute(0, 2, 1, 3) for i in range(3)]      # [B,H,T,d]
    scale = DH ** -0.5
    att = torch.softmax(q @ k.transpose(-1, -2) * scale, dim=-1)
    out = (att @ v).permute(0, 2, 1, 3).reshape(B * T, H * DH)
    dout = rnd(torch.randn(B * T, H * DH, generator=gen), dtype)
    out.backward(dout.double())
    qd = qkv.detach().reshape(B * T, 3 * H * DH).to(DEV).to(dtype)
    od, lse = ops.attention_fwd(qd, B, T, H, DH, scale)
    t = 2e-5 if dtype == torch.float32 else 2e-2
    assert relmax(od.float(), out.detach()) < t
    ref_lse = torch.logsumexp(q @ k.transpose(-1, -2) * scale, dim=-1)
    assert float((lse.cpu().double() - ref_lse).abs().max()) < 1e-4
    dq = ops.attention_bwd(qd, od, dout.to(DEV).to(dtype), lse, B, T, H, DH, scale)
    assert relmax(dq.float().reshape(B, T, 3, H, DH), qkv.grad) < (1e-4 if dtype == torch.float32 else 3e-2)


def test_attention_rejects_unsupported_shapes():
    from passl_amd.hip.lib import PasslHipError
    x = torch.zeros(2 * 300, 3 * 2 * 64, device=DEV)
    with pytest.raises(PasslHipError):
        ops.attention_fwd(x, 2, 300, 2, 64, 0.125)           # T > 208
    with pytest.raises(PasslHipError):
        ops.attention_fwd(torch.zeros(2 * 8, 3 * 2 * 48, device=DEV), 2, 8, 2, 48, 0.1)   # d = 48


def test_masking_and_token_plumbing():
    gen = torch.Generator().manual_seed(2)
    B, L, D, Dd = 5, 196, 64, 32
    noise = torch.rand(B, L, generator=gen)
    noise[0, 3] = noise[0, 100]                                # a tie: lower index first (stable)
    keep, mask, restore = M.random_masking_ids(noise, 0.75)
    K = keep.shape[1]
    ik, ir, mk = ops.mae_mask(noise.to(DEV), K)
    # argsort is not stable for the tie: compare through the defining properties
    assert torch.equal(ir.cpu().long()[1:], restore[1:]) and torch.equal(ik.cpu().long()[1:], keep[1:])
    assert torch.equal(mk.cpu()[1:], mask[1:])
    assert sorted(ir[0].cpu().tolist()) == list(range(L)) and int(ir[0, 3]) + 1 == int(ir[0, 100])
    for dtype in DTYPES:
        x = rnd(torch.randn(B, L, D, generator=gen), dtype)
        cls = torch.randn(D, generator=gen)
        pos = torch.randn(L + 1, D, generator=gen)
        ref = torch.cat([(cls + pos[0]).expand(B, 1, D),
                         torch.gather(x + pos[1:], 1, keep.unsqueeze(-1).expand(-1, -1, D))], 1)
        got = ops.mae_gather(x.reshape(B * L, D).to(DEV).to(dtype), cls.to(DEV), pos.to(DEV), ik, B, L)
        assert relmax(got.float().reshape(B, K + 1, D), ref) < (1e-6 if dtype == torch.float32 else 1e-2)
        dout = rnd(torch.randn(B, K + 1, D, generator=gen), dtype)
        dcls = torch.zeros(D, device=DEV)
        dx = ops.mae_gather_bwd(dout.reshape(-1, D).to(DEV).to(dtype), ir, dcls, B, L, K)
        dref = torch.zeros(B, L, D).scatter_(1, keep.unsqueeze(-1).expand(-1, -1, D), dout[:, 1:])
        assert relmax(dx.float().reshape(B, L, D), dref) < 1e-6 and relmax(dcls, dout[:, 0].sum(0)) < 1e-5
        # decoder side
        xd = rnd(torch.randn(B, K + 1, Dd, generator=gen), dtype)
        mt = torch.randn(Dd, generator=gen)
        dpos = torch.randn(L + 1, Dd, generator=gen)
        x_ = torch.cat([xd[:, 1:], mt.expand(B, L - K, Dd)], 1)
        x_ = torch.gather(x_, 1, restore.unsqueeze(-1).expand(-1, -1, Dd))
        refd = torch.cat([xd[:, :1], x_], 1) + dpos
        gotd = ops.mae_unshuffle(xd.reshape(-1, Dd).to(DEV).to(dtype), mt.to(DEV), dpos.to(DEV), ir, B, K)
        assert relmax(gotd.float().reshape(B, L + 1, Dd), refd) < (1e-6 if dtype == torch.float32 else 1e-2)
        dd = rnd(torch.randn(B, L + 1, Dd, generator=gen), dtype)
        dmt = torch.zeros(Dd, device=DEV)
        dxd = ops.mae_unshuffle_bwd(dd.reshape(-1, Dd).to(DEV).to(dtype), ik, ir, dmt, B)
        dref2 = torch.cat([dd[:, :1], torch.gather(dd[:, 1:], 1, keep.unsqueeze(-1).expand(-1, -1, Dd))], 1)
        assert relmax(dxd.float().reshape(B, K + 1, Dd), dref2) < 1e-6
        assert relmax(dmt, (dd[:, 1:] * mask.unsqueeze(-1)).sum((0, 1))) < 1e-4


def test_patchify_and_masked_patch_loss():
    gen = torch.Generator().manual_seed(3)
    B, p, HW = 3, 16, 64
    img = torch.randn(B, 3, HW, HW, generator=gen)
    ref = M.patchify(img, p)
    for dtype in DTYPES:
        got = ops.patchify(img.to(DEV), p, dtype)
        assert float((got.float().cpu().reshape(ref.shape) - rnd(ref, dtype)).abs().max()) == 0
    L = (HW // p) ** 2
    mask = (torch.rand(B, L, generator=gen) > 0.25).float()
    for norm_pix in (False, True):
        pred = torch.randn(B, L + 1, p * p * 3, generator=gen, dtype=torch.float64).requires_grad_(True)
        tgt = ref.double()
        if norm_pix:
            tgt = (tgt - tgt.mean(-1, keepdim=True)) / (tgt.var(-1, keepdim=True) + 1e-6) ** .5
        loss = ((((pred[:, 1:] - tgt) ** 2).mean(-1)) * mask).sum() / mask.sum()
        (loss * 1.7).backward()
        pd = pred.detach().float().to(DEV).reshape(-1, p * p * 3)
        got = ops.mae_loss_fwd(img.to(DEV), pd, mask.to(DEV), p, norm_pix, float(mask.sum()))
        assert abs(float(got) - float(loss)) < 2e-5 * max(1.0, float(loss))
        dp = ops.mae_loss_bwd(img.to(DEV), pd, mask.to(DEV), torch.tensor([1.7], device=DEV), p, norm_pix,
                              float(mask.sum()))
        assert relmax(dp.reshape(B, L + 1, -1), pred.grad) < 1e-4
        assert float(dp.reshape(B, L + 1, -1)[:, 0].abs().max()) == 0


def test_adamw_kernel_vs_oracle_rule():
    gen = torch.Generator().manual_seed(4)
    n = 100003
    p = torch.randn(n, generator=gen); g = torch.randn(n, generator=gen) * 0.1
    m = torch.zeros(n); v = torch.zeros(n)
    pd, gd, md, vd = p.to(DEV), g.to(DEV), m.to(DEV), v.to(DEV)
    lr, b1, b2, eps, wd = 1e-2, 0.9, 0.95, 1e-8, 0.05
    pr = p.double().clone(); mr = m.double().clone(); vr = v.double().clone()
    for t in (1, 2, 3):
        ops.adamw(pd, gd, md, vd, lr, b1, b2, eps, wd, b1 ** t, b2 ** t, 0.5)
        gg = g.double() * 0.5
        pr = pr * (1 - lr * wd)
        mr = b1 * mr + (1 - b1) * gg
        vr = b2 * vr + (1 - b2) * gg * gg
        pr = pr - lr * math.sqrt(1 - b2 ** t) / (1 - b1 ** t) * mr / (vr.sqrt() + eps * math.sqrt(1 - b2 ** t))
    assert float((pd.cpu().double() - pr).abs().max()) < 1e-5
    assert float((md.cpu().double() - mr).abs().max()) < 1e-6 and float((vd.cpu().double() - vr).abs().max()) < 1e-6


SMALL = dict(img_size=64, patch_size=16, embed_dim=128, depth=2, num_heads=4, decoder_embed_dim=64,
             decoder_depth=2, decoder_num_heads=2, mlp_ratio=4.0)
WATCH = ['patch_embed.proj.weight', 'cls_token', 'mask_token', 'blocks.0.attn.qkv.weight',
         'blocks.1.mlp.fc2.bias', 'blocks.1.norm2.weight', 'norm.bias', 'decoder_embed.weight',
         'decoder_blocks.0.attn.proj.weight', 'decoder_blocks.1.mlp.fc1.weight', 'decoder_pred.bias']
TOL_F32 = dict(loss=1e-3, pred=1e-3, grad=1e-3, param=1e-4)
# bf16 storage of activations / Linear operands (fp32 accumulate, fp32 loss): stated bounds for the
# benchmark dtype (the reference has no bf16 path)
TOL_BF16 = dict(loss=3e-2, pred=1e-1, grad=1e-1, param=2e-2)


def _run_against_golden(name, cfg, dtype, steps_cap, tol):
    z = np.load(os.path.join(GOLDEN, name + '.npz'))
    N, steps, npl = [int(v) for v in z['meta']]
    oracle0 = M.MAEOracle(dict(cfg, norm_pix_loss=bool(npl)), seed=0, **U.SOLVER)
    model, opt = U.build_product(cfg, dtype, bool(npl))
    U.load_oracle_state(model, oracle0)
    model.train()
    gen = torch.Generator().manual_seed(777)
    L = (cfg['img_size'] // cfg['patch_size']) ** 2
    report, bad = [], []

    def check(what, got, ref, nominal, rel=False):
        got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
        scale = max(float(np.max(np.abs(ref))), 1e-12) if rel else 1.0
        err = float(np.max(np.abs(got - ref))) / scale
        line = '%-50s err %.3e  bound %.1e' % (what, err, nominal)
        report.append(line)
        if not err <= nominal:
            bad.append(line)

    for s in range(min(steps, steps_cap)):
        x = torch.randn(N, 3, cfg['img_size'], cfg['img_size'], generator=gen)
        noise = torch.rand(N, L, generator=gen)
        captured = {}
        bb = model.backbone
        orig = bb.forward_loss

        def spy(imgs, pred_rows, mask):
            captured.update(pred=pred_rows.detach(), mask=mask.detach())
            return orig(imgs, pred_rows, mask)
        bb.forward_loss = spy
        out = U.product_step(model, opt, x.to(DEV), noise.to(DEV))
        bb.forward_loss = orig
        pre = 's%d_' % s
        check(pre + 'loss', float(out['loss'].detach()), z[pre + 'loss'], tol['loss'])
        assert np.array_equal(captured['mask'].cpu().numpy().astype(np.uint8), z[pre + 'mask'])
        pred = captured['pred'].float().cpu().reshape(N, L + 1, -1)[:, 1:]
        check(pre + 'pred[:, :4, :8]', pred[:, :4, :8].numpy(), z[pre + 'pred_head'], tol['pred'])
        ps = dict(model.backbone.named_parameters())
        for n in WATCH:
            check(pre + 'gradnorm/' + n, ps[n].grad.double().norm().item(), z[pre + 'gradnorm/' + n],
                  tol['grad'], rel=True)
            check(pre + 'pnorm/' + n, ps[n].detach().double().norm().item(), z[pre + 'pnorm/' + n],
                  tol['param'], rel=True)
    print('\n'.join(report))
    try:
        os.makedirs('gpurun_out', exist_ok=True)
        with open('gpurun_out/parity_%s_%s.txt' % (name, str(dtype).split('.')[-1]), 'w') as f:
            f.write('\n'.join(report) + '\n\nVIOLATIONS (%d)\n' % len(bad) + '\n'.join(bad) + '\n')
    except OSError:
        pass
    assert not bad, 'parity violations:\n' + '\n'.join(bad)


def test_golden_small_fp32():
    _run_against_golden('mae_small', SMALL, torch.float32, 3, TOL_F32)


def test_golden_small_rawpix_fp32():
    _run_against_golden('mae_small_rawpix', SMALL, torch.float32, 1, TOL_F32)


def test_golden_vit_b_fp32():
    """BASELINE configs[3] architecture: ViT-B/16, 50 encoder / 197 decoder tokens."""
    _run_against_golden('mae_vit_b', M.VIT_B, torch.float32, 2, TOL_F32)


def test_golden_small_bf16():
    _run_against_golden('mae_small', SMALL, torch.bfloat16, 3, TOL_BF16)


def test_golden_vit_b_bf16():
    _run_against_golden('mae_vit_b', M.VIT_B, torch.bfloat16, 2, TOL_BF16)


def test_trainer_runs_mae_config_end_to_end(tmp_path):
    """The v110 Trainer + hook bus drive the MAE config (fixed MAE_PRETRAIN wrapper, AdamW,
    LinearWarmup o CosineAnnealingDecay) on synthetic data; the loss goes down."""
    from passl_amd.engine.trainer import Trainer
    from passl_amd.utils.config import get_config
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = get_config(os.path.join(root, 'configs/mae/mae_vit_b_synthetic.yaml'),
                     ['dataloader.train.sampler.batch_size=8', 'dataloader.train.dataset.image_size=64',
                      'dataloader.train.dataset.num_samples=64', 'epochs=3', 'model.architecture.depth=2',
                      'model.architecture.embed_dim=128', 'model.architecture.num_heads=4',
                      'model.architecture.decoder_embed_dim=64', 'model.architecture.decoder_depth=1',
                      'model.architecture.decoder_num_heads=2', 'lr_scheduler.warmup_steps=1',
                      'lr_scheduler.learning_rate.learning_rate=2e-3', 'lr_scheduler.end_lr=2e-3',
                      'output_dir=%s' % tmp_path, 'log_config.interval=4'])
    cfg.model.architecture.img_size = 64
    cfg.timestamp = ''
    tr = Trainer(cfg)
    assert tr.optimizer.type == 'adamw' and tr.iters_per_epoch == 8 and tr.lr_scheduler.warmup_steps == 8
    data = next(iter(tr.train_dataloader))
    tr.model.train()
    l0 = float(tr.model(*data)['loss'].detach())
    tr.train()
    assert tr.current_iter == 24
    l1 = float(tr.outputs['loss'].detach())
    assert np.isfinite(l1) and l1 < l0, (l0, l1)           # one cached batch: it must be fitted


def test_step_is_bit_reproducible():
    """No floating-point atomics on the MAE path (LayerNorm d-gamma / d-beta, class- and mask-token gradients and
    the masked-patch loss are fixed-order slab reductions): two runs from the same state end bit-identical."""
    ends = []
    for _ in range(2):
        oracle = M.MAEOracle(dict(SMALL, norm_pix_loss=True), seed=0, **U.SOLVER)
        model, opt = U.build_product(SMALL, torch.bfloat16, True)
        U.load_oracle_state(model, oracle)
        model.train()
        gen = torch.Generator().manual_seed(31)
        losses = []
        for _s in range(3):
            x = torch.randn(16, 3, 64, 64, generator=gen).to(DEV)
            noise = torch.rand(16, 16, generator=gen).to(DEV)
            losses.append(U.product_step(model, opt, x, noise)['loss'].detach().clone())
        ends.append((torch.cat([l.reshape(1) for l in losses]), model.arena_q.flat.clone()))
    for a, b in zip(*ends):
        assert torch.equal(a, b)


# ---- fine-tuning model: configs/mae/mae_vit_b_finetune.yaml (round-5 verdict, missing #2)
FT_SOLVER = dict(lr=1e-3, beta1=0.9, beta2=0.999, weight_decay=0.05)       # tests/golden/make_golden_mae_finetune.py
FT_ARCH = dict(name='MAE_ViT', patch_size=16, embed_dim=768, depth=12, num_heads=12, qkv_bias=True, mlp_ratio=4)
FT_WATCH = ['backbone.cls_token', 'backbone.pos_embed', 'backbone.patch_embed.proj.weight',
            'backbone.blocks.0.attn.qkv.weight', 'backbone.blocks.1.mlp.fc2.bias', 'backbone.blocks.1.norm2.weight',
            'backbone.fc_norm.weight', 'backbone.fc_norm.bias', 'head.fc_cls.weight', 'head.fc_cls.bias']


def _run_finetune_golden(name, arch, dtype, tol):
    """MAE_FINETUNE over MAE_ViT + VisionTransformerClsHead against the reference's own sources run on torch-CPU
    (tests/golden/<name>.npz).  Step 0 is compared at the nominal bounds; later steps follow Adam updates whose sign is
    rounding-defined wherever a gradient is ~0 (see ADAM_NOISE_DEFINED in test_clip_gpu.py): loss only, x 20."""
    from oracle.mae import finetune_state
    from passl_amd.hip import config as hip_config
    from passl_amd.modeling import build_model
    from passl_amd.solver.optimizer import AdamW
    z = np.load(os.path.join(GOLDEN, name + '.npz'))
    N, hw, steps, classes = [int(v) for v in z['meta']]
    hip_config.set_device('gpu')
    hip_config.set_compute_dtype(dtype)
    torch.manual_seed(0)
    model = build_model(dict(name='MAE_FINETUNE', architecture=dict(arch),
                             head=dict(name='VisionTransformerClsHead', num_classes=classes,
                                       in_channels=arch['embed_dim'])))
    sd = model.state_dict()
    keys_shapes = [(k, tuple(v.shape)) for k, v in sd.items()]
    assert ['%s:%s' % (k, 'x'.join(map(str, s))) for k, s in keys_shapes] == [str(k) for k in z['keys']]
    missing, unexpected = model.load_state_dict({k: v for k, v in finetune_state(keys_shapes).items()}, strict=False)
    assert not missing and not unexpected
    model.train()
    opt = AdamW(FT_SOLVER['lr'], beta1=FT_SOLVER['beta1'], beta2=FT_SOLVER['beta2'],
                weight_decay=FT_SOLVER['weight_decay'], parameters=list(model.parameters()))
    seen = {}
    head_fwd = model.head.forward

    def spy(x):
        seen['feat'] = x.detach()
        seen['score'] = head_fwd(x)
        return seen['score']
    model.head.forward = spy
    gen = torch.Generator().manual_seed(909)
    report, bad = [], []

    def check(what, got, ref, bound, rel=False):
        got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
        scale = max(float(np.max(np.abs(ref))), 1e-12) if rel else 1.0
        err = float(np.max(np.abs(got - ref))) / scale
        line = '%-52s err %.3e  bound %.3e' % (what, err, bound)
        report.append(line)
        if not err <= bound:
            bad.append(line)

    for s in range(steps):
        x = torch.randn(N, 3, hw, hw, generator=gen)
        y = torch.randint(0, classes, (N,), generator=gen)
        out = model(x.to(DEV), y.to(DEV), mode='train')
        opt.clear_grad()
        out['loss'].backward()
        pre = 's%d_' % s
        k = 1.0 if s == 0 else 20.0
        check(pre + 'loss', float(out['loss'].detach()), z[pre + 'loss'], tol['loss'] * k, rel=True)
        if s == 0:
            check(pre + 'acc1', float(out['acc1']), z[pre + 'acc1'], 1e-6)
            check(pre + 'acc5', float(out['acc5']), z[pre + 'acc5'], 1e-6)
            check(pre + 'feat[:, :8]', seen['feat'].float().cpu()[:, :8].numpy(), z[pre + 'feat_head'], tol['feat'], rel=True)
            check(pre + 'score[:, :8]', seen['score'].detach().float().cpu()[:, :8].numpy(), z[pre + 'score_head'],
                  tol['feat'], rel=True)
            ps = dict(model.named_parameters())
            for n in FT_WATCH:
                check(pre + 'gradnorm/' + n, ps[n].grad.double().norm().item(), z[pre + 'gradnorm/' + n], tol['grad'], rel=True)
        opt.step()
        if s == 0:
            ps = dict(model.named_parameters())
            for n in FT_WATCH:
                check(pre + 'pnorm/' + n, ps[n].detach().double().norm().item(), z[pre + 'pnorm/' + n], tol['param'], rel=True)
    print('\n'.join(report))
    try:
        os.makedirs('gpurun_out', exist_ok=True)
        with open('gpurun_out/parity_%s_%s.txt' % (name, str(dtype).split('.')[-1]), 'w') as f:
            f.write('\n'.join(report) + '\n\nVIOLATIONS (%d)\n' % len(bad) + '\n'.join(bad) + '\n')
    except OSError:
        pass
    assert not bad, 'parity violations:\n' + '\n'.join(bad)


FT_TOL_F32 = dict(loss=1e-3, feat=1e-3, grad=2e-3, param=1e-4)
FT_TOL_BF16 = dict(loss=3e-2, feat=6e-2, grad=8e-2, param=1e-2)      # the CLIP / MAE bf16 bounds (stated, not parity)


def test_finetune_golden_small_fp32():
    _run_finetune_golden('mae_ft_small', dict(FT_ARCH, embed_dim=128, depth=2, num_heads=4, img_size=64),
                         torch.float32, FT_TOL_F32)


def test_finetune_golden_vit_b_fp32():
    _run_finetune_golden('mae_ft_vit_b', dict(FT_ARCH), torch.float32, FT_TOL_F32)


def test_finetune_golden_vit_b_bf16():
    _run_finetune_golden('mae_ft_vit_b', dict(FT_ARCH), torch.bfloat16, FT_TOL_BF16)


def test_trainer_runs_mae_finetune_config_end_to_end(tmp_path):
    """configs/mae/mae_vit_b_finetune_synthetic.yaml (the reference YAML's model / lr / optimizer blocks over synthetic
    labelled batches) through the v110 Trainer + hook bus for a few iterations."""
    from passl_amd.engine.trainer import Trainer
    from passl_amd.utils.config import get_config
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = get_config(os.path.join(root, 'configs/mae/mae_vit_b_finetune_synthetic.yaml'),
                     ['dataloader.train.sampler.batch_size=8', 'dataloader.train.dataset.num_samples=32', 'epochs=1',
                      'output_dir=%s' % tmp_path, 'log_config.interval=2'])
    cfg.timestamp = ''
    tr = Trainer(cfg)
    assert type(tr.model).__name__ == 'MAE_FINETUNE' and tr.iters_per_epoch == 4 and tr.optimizer.type == 'adamw'
    w0 = tr.model.head.fc_cls.weight.detach().clone()
    tr.train()
    assert tr.current_iter == 4
    loss = float(tr.outputs['loss'].detach())
    assert np.isfinite(loss) and 0 < loss < 20 and 'acc1' in tr.outputs
    assert float((tr.model.head.fc_cls.weight.detach() - w0).abs().max()) > 0
