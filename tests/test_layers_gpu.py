"""Per-layer, TEACHER-FORCED bf16 parity of the ResNet-50 hot path on a real MI355X.

End-to-end comparison through a random-init network with batch-statistics BatchNorm is chaotic (two valid bf16
evaluations drift apart layer by layer), so it cannot bound the bf16 path tightly.  Here every layer is checked
on its own: the bf16-emulating oracle (oracle/resnet50.py, ``rec=``: the reference's algorithm with bfloat16
rounding at the product's storage points, pinned by the reference's own sources) evaluates the trunk once on
the CPU and keeps EVERY stored tensor and its gradient; each product layer then receives the ORACLE's input
(forward) and the ORACLE's output gradient (backward) and must reproduce the oracle's output / input gradient /
parameter gradients:

  unit A  conv (+ fused BatchNorm statistics in its epilogue) -> BatchNorm (+ residual) (+ ReLU), then backward:
          BatchNorm backward (reduce + apply), data gradient, weight gradient
  unit B  BatchNorm -> its sole consumer conv, backward only: the data-gradient launch carries the fused
          BatchNorm-backward reduction (bf16-only path that no fp32 whole-step test executes)
  unit C  stem max-pool forward / backward, SimCLR projector (Linear with fp32 output -> BatchNorm1D -> ReLU)

Bounds.  bf16 tensors: max|d| <= 2 ulp and mean|d| <= 1/2 ulp, ulp = the bf16 spacing at the reference tensor's
largest magnitude (both sides round an fp32 accumulator that differs by summation order only, so single elements
flip by one rounding step).  fp32 results (weight / affine gradients, running statistics): relative to the
reference's largest magnitude.  Geometries: MoCo cfg-2 (224^2, stem max-pool) and SimCLR (224^2, pool-free trunk:
four times the rows per layer) at a batch the CPU oracle evaluates in seconds.
"""
import math
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

import moco_util as U                                   # noqa: E402
import simclr_util as SU                                # noqa: E402
from oracle import resnet50 as R                        # noqa: E402
from oracle import simclr as S                          # noqa: E402
from oracle.bf16 import _rne                            # noqa: E402
from oracle.moco import MoCoOracle                      # noqa: E402
from passl_amd.hip import config as hip_config, nn, ops  # noqa: E402

DEV = 'cuda'


def ulp_bf16(maxabs):
    return 2.0 ** (math.floor(math.log2(max(float(maxabs), 1e-30))) - 7)


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous() if t.dim() == 4 else t


class Report:
    def __init__(self, name):
        self.name, self.lines, self.bad = name, [], []

    def bf16(self, what, got, ref, max_ulps=2.0, mean_ulps=0.5, relu_of=None):
        """got: HIP tensor (NHWC / rows), ref: oracle tensor holding bf16 values (NCHW / rows).
        relu_of = (z_hip, z_oracle): outputs of the ReLU this gradient passes through.  Where the two sides
        disagree about z > 0 — the pre-activation is zero to within rounding, so either answer is a valid
        evaluation — the element is left out of the comparison; at most 0.01 % of a tensor may be left out."""
        ref = nhwc(ref.detach().float())
        got = got.detach().float().cpu()
        assert got.shape == ref.shape, (what, got.shape, ref.shape)
        u = ulp_bf16(ref.abs().max())
        d = (got - ref).abs()
        note = ''
        if relu_of is not None:
            zh, zo = relu_of
            flip = (zh.detach().float().cpu() > 0) != (nhwc(zo.detach().float()) > 0)
            nflip = int(flip.sum())
            d = torch.where(flip, torch.zeros_like(d), d)
            note = '  [%d ReLU-boundary elements left out]' % nflip
            if nflip > 1e-4 * d.numel():
                self.bad.append('%s: %d of %d elements disagree about the ReLU mask' % (what, nflip, d.numel()))
        mx, mean, frac = float(d.max()) / u, float(d.mean()) / u, float((d > 0).float().mean())
        line = '%-44s max %.2f ulp  mean %.4f ulp  differing %.3f %%  (ulp %.3e, |ref|max %.3e)%s' % (
            what, mx, mean, 100 * frac, u, float(ref.abs().max()), note)
        self.lines.append(line)
        if not (mx <= max_ulps and mean <= mean_ulps) or not torch.isfinite(got).all():
            self.bad.append(line)

    def f32(self, what, got, ref, rel, allow=None):
        """allow: per-element absolute allowance on top of rel * max|ref| (see relu_allowance)."""
        ref = ref.detach().double()
        got = got.detach().double().cpu().reshape(ref.shape)
        scale = max(float(ref.abs().max()), 1e-30)
        d = (got - ref).abs()
        note = ''
        if allow is not None:
            allow = allow.double().reshape(ref.shape)
            d = (d - allow).clamp_min(0.0)
            note = '  [ReLU-boundary allowance up to %.2e]' % float(allow.max())
        err = float(d.max()) / scale
        line = '%-44s rel-to-max err %.3e  (bound %.1e, |ref|max %.3e)%s' % (what, err, rel, scale, note)
        self.lines.append(line)
        if not err <= rel:
            self.bad.append(line)

    def finish(self):
        print('\n'.join(self.lines))
        try:
            os.makedirs('gpurun_out', exist_ok=True)
            with open('gpurun_out/parity_layers_%s.txt' % self.name, 'w') as f:
                f.write('\n'.join(self.lines) + '\n\nVIOLATIONS (%d)\n' % len(self.bad) + '\n'.join(self.bad) + '\n')
        except OSError:
            pass
        assert not self.bad, 'per-layer parity violations:\n' + '\n'.join(self.bad)


def _oracle_run(geometry, N, hw, seed):
    """One training-mode forward + backward of the bf16-emulating encoder on the CPU, every stored tensor kept.
    moco_cfg2: MoCo-v2's query trunk (stem max-pool, kaiming init); simclr: SimCLR's pool-free trunk (Xavier init)
    and its projector (NonLinearNeckfc3)."""
    gen = torch.Generator().manual_seed(seed + 100)
    img = torch.randn(N, 3, hw, hw, generator=gen)
    rec, new_stats = {}, {}
    if geometry == 'moco_cfg2':
        oracle = MoCoOracle(K=256, seed=seed, bf16=True)
        st = oracle.q
        for n in R.trainable_keys(st):
            st[n] = st[n].detach().requires_grad_(True)
        y = R.trunk_forward(st, img, False, new_stats, None, maxpool=True, bf16=True, rec=rec)
    else:
        oracle = S.SimCLROracle(seed=seed, bf16=True)
        st = oracle.st
        for n in R.trainable_keys(st):
            st[n] = st[n].detach().requires_grad_(True)
        y = S.encoder_forward(st, img, new_stats, None, bf16=True, rec=rec)
    # any well-conditioned scalar: a fixed random projection of the output (gradients of every sign and size)
    proj = torch.randn(y.shape, generator=gen)
    (y * proj).sum().backward()
    return oracle, st, new_stats, rec, img


def relu_allowance(z_hip, z_ref, dz_ref, y_ref):
    """BatchNorm affine gradients are sums over all positions of the MASKED output gradient g (dbeta = sum g,
    dgamma = sum g * xhat).  Where the two sides disagree about z > 0 (pre-activation zero to within rounding: both
    answers are valid evaluations) one side's sum contains dz and the other's does not: per channel the sums may
    differ by sum |dz| resp. sum |dz * xhat| over those positions.  Returns (allow_dgamma[C], allow_dbeta[C])."""
    zr = z_ref.detach().float()
    flip = ((nhwc(zr) > 0) != (z_hip.detach().float().cpu() > 0)).float()          # NHWC / rows
    dz = nhwc(dz_ref.detach().float()).abs() * flip
    yr = y_ref.detach().float()
    dims = (0, 2, 3) if yr.dim() == 4 else (0,)
    mean = yr.mean(dim=dims, keepdim=True)
    inv = torch.rsqrt(yr.var(dim=dims, unbiased=False, keepdim=True) + R.BN_EPS)
    xhat = nhwc(((yr - mean) * inv)).abs()
    red = tuple(range(dz.dim() - 1))
    return (dz * xhat).sum(dim=red), dz.sum(dim=red)


def _dev(t, requires_grad=False):
    """oracle tensor (bf16 values in fp32, NCHW) -> NHWC bf16 device tensor"""
    d = nhwc(t.detach()).to(DEV).to(torch.bfloat16)
    return d.requires_grad_(True) if requires_grad else d


def _grad_bf16(t):
    """the gradient a backward kernel reads: the fp32 gradient w.r.t. a stored tensor, rounded (oracle/bf16.py)"""
    return _rne(t.grad)


def _layer(backbone, name):
    m = backbone
    for part in name.split('.'):
        m = m[int(part)] if part.isdigit() else getattr(m, part)
    return m


def _reset_bn(bn):
    with torch.no_grad():
        bn._mean.zero_()
        bn._variance.fill_(1.0)


@pytest.mark.parametrize('geometry', ['moco_cfg2', 'simclr'])
def test_r50_layers_teacher_forced_bf16(geometry):
    maxpool = geometry == 'moco_cfg2'
    N = 8 if maxpool else 4
    oracle, st, new_stats, rec, img = _oracle_run(geometry, N, 224, seed=3)
    if maxpool:
        model, _opt, _sched = U.build_product(256, torch.bfloat16)
        U.load_oracle_state(model, oracle)
        backbone, neck = model.encoder_q[0], None
    else:
        model, _opt, _sched = SU.build_product(torch.bfloat16)
        SU.load_oracle_state(model, oracle)
        backbone, neck = model.encoder[0], model.encoder[1]
    model.train()
    arena = backbone.conv1._rt.arena
    assert hip_config.fused_bn_stats() and hip_config.fused_bn_backward()
    rep = Report('r50_%s_bfloat16' % geometry)
    pre = '0.'                                  # state keys of the backbone inside nn.Sequential(backbone, neck)
    # spy: which launches carried the fused BatchNorm-backward epilogue / the fused statistics
    seen = dict(bnb=0, stats=0)
    real_conv = ops.conv_igemm

    def spy(d, a, b, y, **kw):
        seen['bnb'] += kw.get('bnb') is not None
        seen['stats'] += kw.get('stats') is not None
        return real_conv(d, a, b, y, **kw)
    ops.conv_igemm = spy
    try:
        # ------------------------------------------------------------------ unit A: every conv + its BatchNorm
        for name, _cout, _cin, _k, _s, _p, bn_name in R.conv_specs():
            conv, bn = _layer(backbone, name), _layer(backbone, bn_name)
            relu = not bn_name.endswith('downsample.1')
            _reset_bn(bn)
            arena.clear_grad()
            is_stem = name == 'conv1'
            if is_stem:
                xp, H, W = backbone._stem_input(img.to(DEV))
                y, stats = conv(xp, hw=(H, W), want_stats=True)
                xh = None
            else:
                xh = _dev(rec[name + '.x'], requires_grad=True)
                y, stats = conv(xh, want_stats=True)
            assert stats is not None
            # the BatchNorm is taught separately from its conv: it sees the conv's output values (and the fused
            # statistics of exactly these values) as a leaf, so that each backward gets the ORACLE's gradient
            yl = y.detach().requires_grad_(True)
            res = _dev(rec[bn_name + '.res']) if (bn_name + '.res') in rec else None
            z = bn(yl, residual=res, relu=relu, stats=stats)
            rep.bf16(name + ' fwd (+stats)', y, rec[name + '.y'])
            rep.bf16(bn_name + ' fwd', z, rec[bn_name + '.z'])
            rep.f32(bn_name + ' running mean', bn._mean, new_stats[pre + bn_name + '._mean'], 2e-3)
            rep.f32(bn_name + ' running var', bn._variance, new_stats[pre + bn_name + '._variance'], 2e-3)
            z.backward(_dev(_grad_bf16(rec[bn_name + '.z'])))
            y.backward(_dev(_grad_bf16(rec[name + '.y'])))
            torch.cuda.synchronize()
            rep.bf16(bn_name + ' bwd -> d(conv out)', yl.grad, _grad_bf16(rec[name + '.y']),
                     relu_of=(z, rec[bn_name + '.z']) if relu else None)
            if xh is not None:
                rep.bf16(name + ' dgrad', xh.grad, _grad_bf16(rec[name + '.x']))
            rep.f32(name + ' wgrad', conv.weight.grad, st[pre + name + '.weight'].grad, 2e-3)
            ag, ab = relu_allowance(z, rec[bn_name + '.z'], _grad_bf16(rec[bn_name + '.z']), rec[name + '.y']) \
                if relu else (None, None)
            rep.f32(bn_name + ' dgamma', bn.weight.grad, st[pre + bn_name + '.weight'].grad, 5e-3, allow=ag)
            rep.f32(bn_name + ' dbeta', bn.bias.grad, st[pre + bn_name + '.bias'].grad, 5e-3, allow=ab)
        assert seen['stats'] == 53 and seen['bnb'] == 0
        # ------------------------------------------------------------------ unit B: BatchNorm -> sole consumer conv
        n_b = 0
        for li, blocks in enumerate(R.LAYERS, start=1):
            for b in range(blocks):
                p = 'layer%d.%d' % (li, b)
                for prod, cons in ((p + '.conv1', p + '.conv2'), (p + '.conv2', p + '.conv3')):
                    bn_name = prod.replace('conv', 'bn')
                    bn, conv = _layer(backbone, bn_name), _layer(backbone, cons)
                    _reset_bn(bn)
                    arena.clear_grad()
                    yp = _dev(rec[prod + '.y'], requires_grad=True)
                    z = bn(yp, relu=True)                           # statistics by the stand-alone pass
                    link = nn.bn_link(z)
                    assert link is not None
                    before = seen['bnb']
                    yc = conv(z, producer=link)
                    yc.backward(_dev(_grad_bf16(rec[cons + '.y'])))
                    torch.cuda.synchronize()
                    assert seen['bnb'] > before, 'the data-gradient launch did not take the fused BatchNorm backward'
                    rep.bf16('%s -> %s: fused bwd d(conv out)' % (bn_name, cons), yp.grad, _grad_bf16(rec[prod + '.y']),
                             relu_of=(z, rec[bn_name + '.z']))
                    ag, ab = relu_allowance(z, rec[bn_name + '.z'], _grad_bf16(rec[bn_name + '.z']), rec[prod + '.y'])
                    rep.f32('%s (fused) dgamma' % bn_name, bn.weight.grad, st['0.' + bn_name + '.weight'].grad, 5e-3, allow=ag)
                    rep.f32('%s (fused) dbeta' % bn_name, bn.bias.grad, st['0.' + bn_name + '.bias'].grad, 5e-3, allow=ab)
                    rep.f32('%s wgrad (behind fused bwd)' % cons, conv.weight.grad, st['0.' + cons + '.weight'].grad, 2e-3)
                    n_b += 1
        assert n_b == 32
        # ------------------------------------------------------------------ unit C: stem max-pool
        if maxpool:
            xin = _dev(rec['bn1.z'], requires_grad=True)
            zp = backbone.maxpool(xin)
            rep.bf16('maxpool fwd', zp, rec['maxpool.z'])
            zp.backward(_dev(_grad_bf16(rec['maxpool.z'])))
            torch.cuda.synchronize()
            # a window whose maximum is attained more than once (ReLU zeros) may route its gradient to either
            # position; at a zero of the ReLU output that gradient is masked by the BatchNorm backward anyway
            live = (rec['bn1.z'].detach() > 0).float()
            rep.bf16('maxpool bwd (where the input is > 0)', xin.grad * nhwc(live).to(DEV), _grad_bf16(rec['bn1.z']) * live)
        # ------------------------------------------------------------------ unit C: SimCLR projector
        if neck is not None:
            for i_fc, i_bn, relu in ((0, 1, True), (3, 4, True), (6, 7, False)):
                fc, bn = neck.mlp[i_fc], neck.mlp[i_bn]
                _reset_bn(bn)
                arena.clear_grad()
                xh = _dev(rec['mlp.%d.x' % i_fc], requires_grad=True)
                y = fc(xh, out_f32=True)
                y.retain_grad()
                z = bn(y, relu=relu)
                rep.f32('neck mlp.%d (Linear, fp32 out) fwd' % i_fc, y, rec['mlp.%d.y' % i_fc], 1e-4)
                rep.f32('neck mlp.%d (BatchNorm1D) fwd' % i_bn, z, rec['mlp.%d.z' % i_bn], 1e-3)
                z.backward(rec['mlp.%d.z' % i_bn].grad.to(DEV))
                torch.cuda.synchronize()
                rep.f32('neck mlp.%d bwd -> d(fc out)' % i_bn, y.grad, rec['mlp.%d.y' % i_fc].grad, 1e-3)
                rep.bf16('neck mlp.%d dgrad' % i_fc, xh.grad, _grad_bf16(rec['mlp.%d.x' % i_fc]))
                rep.f32('neck mlp.%d wgrad' % i_fc, fc.weight.grad, st['1.mlp.%d.weight' % i_fc].grad, 2e-3)
                # (the Linear's bias feeds a batch-statistics BatchNorm: its gradient is mathematically zero, both
                # sides hold rounding residue of size 1e-6 — nothing to compare)
                rep.f32('neck mlp.%d dgamma' % i_bn, bn.weight.grad, st['1.mlp.%d.weight' % i_bn].grad, 5e-3)
                rep.f32('neck mlp.%d dbeta' % i_bn, bn.bias.grad, st['1.mlp.%d.bias' % i_bn].grad, 5e-3)
    finally:
        ops.conv_igemm = real_conv
    rep.finish()


# ======================================================================================== ViT block
def _rb(t):
    """Round to bfloat16 (RNE), keep as fp64 for the next exact evaluation."""
    return t.float().bfloat16().double()


def _vit_block_oracle(B, T, D, H, seed, quick_gelu=False):
    """One pre-norm transformer block (reference passl/models/vision_transformer.py:159-249 = passl_v110/
    modeling/backbones/mae.py:61-189) evaluated OP BY OP in fp64 from bfloat16-valued inputs, the result of every
    op rounded to bfloat16 where the product stores bf16 (the input of the next op is that rounded tensor):
    forward tensors and, from a bfloat16-valued output gradient, every op's input gradient (rounded likewise) and
    parameter gradients (fp64).  Weights: Xavier-uniform fp32 masters; the GEMM operands are their bf16 roundings."""
    gen = torch.Generator().manual_seed(seed)
    hid, dh = 4 * D, D // H

    def xav(i, o):
        return ((torch.rand(i, o, generator=gen) * 2 - 1) * math.sqrt(6.0 / (i + o)))
    P = {'norm1.weight': 1 + 0.1 * torch.randn(D, generator=gen), 'norm1.bias': 0.1 * torch.randn(D, generator=gen),
         'attn.qkv.weight': xav(D, 3 * D), 'attn.qkv.bias': 0.02 * torch.randn(3 * D, generator=gen),
         'attn.proj.weight': xav(D, D), 'attn.proj.bias': 0.02 * torch.randn(D, generator=gen),
         'norm2.weight': 1 + 0.1 * torch.randn(D, generator=gen), 'norm2.bias': 0.1 * torch.randn(D, generator=gen),
         'mlp.fc1.weight': xav(D, hid), 'mlp.fc1.bias': 0.02 * torch.randn(hid, generator=gen),
         'mlp.fc2.weight': xav(hid, D), 'mlp.fc2.bias': 0.02 * torch.randn(D, generator=gen)}
    rec, grads = {}, {}
    x0 = _rb(torch.randn(B * T, D, generator=gen) * 1.5)
    dx2 = _rb(torch.randn(B * T, D, generator=gen) * 1e-3)

    def leaf(t):
        return t.detach().clone().requires_grad_(True)

    def op(name, fn, inputs, dout, params=()):
        """Evaluate fn on leaf copies of `inputs` (bf16-valued fp64) and the fp64 params; returns the bf16-rounded
        output.  With `dout` (bf16-valued): input gradients (rounded) and parameter gradients (fp64)."""
        ins = [leaf(t) for t in inputs]
        ps = [leaf(P[k].double()) for k in params]
        out = fn(*ins, *ps)
        rec[name + '.in'] = [t.detach() for t in inputs]
        rec[name + '.out'] = _rb(out.detach())
        if dout is not None:
            out.backward(dout)
            rec[name + '.dout'] = dout
            rec[name + '.din'] = [_rb(t.grad) for t in ins]
            for k, p in zip(params, ps):
                grads[k] = p.grad.clone()
        return rec[name + '.out']

    def ln(x, g, b):
        return torch.nn.functional.layer_norm(x, (D,), g, b, 1e-6)

    def lin(x, w, b, res=None):
        y = x @ w
        return y + b if res is None else y + b + res

    def attn(qkv):
        q, k, v = [qkv.reshape(B, T, 3, H, dh)[:, :, i].permute(0, 2, 1, 3) for i in range(3)]
        a = torch.softmax(q @ k.transpose(-1, -2) * dh ** -0.5, dim=-1)
        return (a @ v).permute(0, 2, 1, 3).reshape(B * T, D)

    def act(x):
        return x * torch.sigmoid(1.702 * x) if quick_gelu else torch.nn.functional.gelu(x)

    # the GEMM kernels read the bf16 rounding of the fp32 master weights
    Pm = dict(P)
    for k in ('attn.qkv.weight', 'attn.proj.weight', 'mlp.fc1.weight', 'mlp.fc2.weight'):
        P[k] = _rb(P[k]).float()
    # ---- forward (outputs recorded), then backward op by op in reverse with the recorded inputs
    fwd = {}
    fwd['h1'] = op('norm1', ln, [x0], None, ('norm1.weight', 'norm1.bias'))
    fwd['qkv'] = op('qkv', lin, [fwd['h1']], None, ('attn.qkv.weight', 'attn.qkv.bias'))
    fwd['a'] = op('attn', attn, [fwd['qkv']], None)
    fwd['x1'] = op('proj', lambda a, r, w, b: lin(a, w, b, r), [fwd['a'], x0], None,
                   ('attn.proj.weight', 'attn.proj.bias'))
    fwd['h2'] = op('norm2', ln, [fwd['x1']], None, ('norm2.weight', 'norm2.bias'))
    fwd['f1'] = op('fc1', lin, [fwd['h2']], None, ('mlp.fc1.weight', 'mlp.fc1.bias'))
    fwd['g'] = op('act', act, [fwd['f1']], None)
    fwd['x2'] = op('fc2', lambda g_, r, w, b: lin(g_, w, b, r), [fwd['g'], fwd['x1']], None,
                   ('mlp.fc2.weight', 'mlp.fc2.bias'))
    op('fc2', lambda g_, r, w, b: lin(g_, w, b, r), [fwd['g'], fwd['x1']], dx2, ('mlp.fc2.weight', 'mlp.fc2.bias'))
    dg, dres2 = rec['fc2.din']
    op('act', act, [fwd['f1']], dg)
    op('fc1', lin, [fwd['h2']], rec['act.din'][0], ('mlp.fc1.weight', 'mlp.fc1.bias'))
    # norm2 is a residual fork: dx1 = LN-backward(dh2) + the skip branch's gradient, added in ONE kernel
    op('norm2', ln, [fwd['x1']], rec['fc1.din'][0], ('norm2.weight', 'norm2.bias'))
    rec['norm2.fork_din'] = _rb(rec['norm2.din'][0] + dres2)
    op('proj', lambda a, r, w, b: lin(a, w, b, r), [fwd['a'], x0], rec['norm2.fork_din'],
       ('attn.proj.weight', 'attn.proj.bias'))
    da, dres1 = rec['proj.din']
    op('attn', attn, [fwd['qkv']], da)
    op('qkv', lin, [fwd['h1']], rec['attn.din'][0], ('attn.qkv.weight', 'attn.qkv.bias'))
    op('norm1', ln, [x0], rec['qkv.din'][0], ('norm1.weight', 'norm1.bias'))
    rec['norm1.fork_din'] = _rb(rec['norm1.din'][0] + dres1)
    return Pm, rec, grads, x0, dx2


@pytest.mark.parametrize('geometry', ['mae_encoder', 'clip_b16'])
def test_vit_block_teacher_forced_bf16(geometry):
    """Every kernel of a ViT-B block (LayerNorm, Linear + bias (+ residual) epilogues, fused attention, GELU /
    QuickGELU) forward and backward, each fed the ORACLE's bfloat16 input / output gradient of that op and held to
    the R50 layers' bound: max|d| <= 2 bf16 ulp of the tensor's largest magnitude, mean|d| <= 1/2 ulp; fp32
    parameter gradients relative to their largest magnitude.  mae_encoder: 50 tokens (MAE's visible set + class
    token), exact GELU; clip_b16: 197 tokens, QuickGELU."""
    from passl_amd.modeling.backbones import mae as MB
    from passl_amd.modeling.backbones import vision_transformer as VB
    quick = geometry == 'clip_b16'
    B, T, D, H = (6, 50, 768, 12) if not quick else (3, 197, 768, 12)
    Pm, rec, grads, x0, dx2 = _vit_block_oracle(B, T, D, H, seed=17, quick_gelu=quick)
    hip_config.set_device('gpu')
    hip_config.set_compute_dtype(torch.bfloat16)
    if quick:
        blk = VB.Block(dim=D, num_heads=H, mlp_ratio=4.0, qkv_bias=True, epsilon=1e-6)
    else:
        from functools import partial
        blk = MB.Block(D, H, 4.0, qkv_bias=True, norm_layer=partial(nn.LayerNorm, epsilon=1e-6))
    arena = nn.EncoderArena(blk, trainable=True)
    missing, unexpected = blk.load_state_dict({k: v.float() for k, v in Pm.items()}, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    arena.refresh()
    rep = Report('vit_block_%s' % geometry)

    def dev(t):
        return t.float().to(DEV).to(torch.bfloat16)

    def run(name, call, n_in, param_keys=(), fork_res=None):
        """Forward with the oracle's inputs, backward with the oracle's output gradient."""
        arena.clear_grad()
        ins = [dev(t).requires_grad_(True) for t in rec[name + '.in'][:n_in]]
        out = call(*ins)
        rep.bf16(name + ' fwd', out, rec[name + '.out'])
        out.backward(dev(rec[name + '.dout']))
        torch.cuda.synchronize()
        for i, t in enumerate(ins):
            ref = rec[name + '.din'][i]
            rep.bf16('%s d-input %d' % (name, i), t.grad, ref)
        ps = dict(blk.named_parameters())
        for k in param_keys:
            rep.f32('%s d %s' % (name, k), ps[k].grad, grads[k], 1e-5)

    attn = blk.attn
    run('norm1', lambda x: blk.norm1(x), 1, ('norm1.weight', 'norm1.bias'))
    run('qkv', lambda h: attn.qkv(h), 1, ('attn.qkv.weight', 'attn.qkv.bias'))
    run('attn', lambda q: nn.attention(q, B, T, attn.num_heads, attn.head_dim, attn.scale), 1)
    run('proj', lambda a, r: attn.proj(a, residual=r), 2, ('attn.proj.weight', 'attn.proj.bias'))
    run('norm2', lambda x: blk.norm2(x), 1, ('norm2.weight', 'norm2.bias'))
    run('fc1', lambda h: blk.mlp.fc1(h), 1, ('mlp.fc1.weight', 'mlp.fc1.bias'))
    run('act', lambda f: blk.mlp.act(f), 1)
    run('fc2', lambda g, r: blk.mlp.fc2(g, residual=r), 2, ('mlp.fc2.weight', 'mlp.fc2.bias'))
    # the residual forks: LayerNorm backward + skip gradient in one kernel (nn.LayerNorm.fork)
    for name, norm, dres in (('norm2', blk.norm2, rec['fc2.din'][1]), ('norm1', blk.norm1, rec['proj.din'][1])):
        arena.clear_grad()
        x = dev(rec[name + '.in'][0]).requires_grad_(True)
        h, xr = norm.fork(x)
        torch.autograd.backward([h, xr], [dev(rec[name + '.dout']), dev(dres)])
        rep.bf16(name + ' fork d-input', x.grad, rec[name + '.fork_din'])
    rep.finish()


# ======================================================================================== BatchNorm-1D MLP heads
def _head_oracle(dims, N, seed, last_affine, bias_last=False):
    """Linear(no bias) - BatchNorm1D - ReLU ... chains of the v2 heads (MoCo-v3 projector / predictor,
    passl/models/mocov3.py:135-157; SimSiam projector / predictor, passl/models/simsiam.py:47-66) evaluated OP BY OP in
    fp64 under the product's storage contract: a Linear reads the bfloat16 rounding of its input rows and of its fp32
    master weight and WRITES fp32; BatchNorm1D (batch statistics, biased variance, eps 1e-5) works on fp32 rows; the
    gradient entering a Linear is rounded to bfloat16, its input gradient is stored as bfloat16.  -> params, rec, grads."""
    gen = torch.Generator().manual_seed(seed)
    P, rec, grads = {}, {}, {}
    n_lin = len(dims) - 1
    for l in range(n_lin):
        i, o = dims[l], dims[l + 1]
        P['lin%d.weight' % l] = ((torch.rand(i, o, generator=gen) * 2 - 1) * math.sqrt(6.0 / (i + o)))
        if bias_last and l == n_lin - 1:
            P['lin%d.bias' % l] = 0.05 * torch.randn(o, generator=gen)
        has_bn = l < n_lin - 1 or not bias_last
        if has_bn and (l < n_lin - 1 or last_affine):
            P['bn%d.weight' % l] = 1 + 0.2 * torch.randn(o, generator=gen)
            P['bn%d.bias' % l] = 0.2 * torch.randn(o, generator=gen)

    def leaf(t):
        return t.detach().clone().requires_grad_(True)

    def bn(y, g=None, b=None):
        mu = y.mean(0, keepdim=True)
        var = ((y - mu) ** 2).mean(0, keepdim=True)
        z = (y - mu) / torch.sqrt(var + 1e-5)
        return z if g is None else z * g + b

    x = torch.randn(N, dims[0], generator=gen).double() * 1.3 + 0.2          # fp32-valued rows
    x = x.float().double()
    fwd_in = x
    chain = []
    for l in range(n_lin):
        wb = _rb(P['lin%d.weight' % l])
        xin = fwd_in
        y = _rb(xin) @ wb
        if 'lin%d.bias' % l in P:
            y = y + P['lin%d.bias' % l].double()
        y = y.float().double()                                            # stored fp32
        rec['lin%d.in' % l], rec['lin%d.out' % l] = xin, y
        has_bn = ('bn%d.weight' % l in P) or (l == n_lin - 1 and not bias_last)
        relu = l < n_lin - 1
        if has_bn or relu:
            g = P.get('bn%d.weight' % l)
            b = P.get('bn%d.bias' % l)
            z = bn(y, g.double() if g is not None else None, b.double() if b is not None else None)
            if relu:
                z = torch.relu(z)
            z = z.float().double()
            rec['bn%d.in' % l], rec['bn%d.out' % l] = y, z
            fwd_in = z
        else:
            fwd_in = y
        chain.append((l, has_bn or relu, relu))
    dout = (torch.randn(N, dims[-1], generator=gen) * 1e-2).float().double()
    for l, has_bn, relu in reversed(chain):
        if has_bn:
            yl = leaf(rec['bn%d.in' % l])
            ps = [leaf(P[k].double()) for k in ('bn%d.weight' % l, 'bn%d.bias' % l) if k in P]
            z = bn(yl, *ps) if ps else bn(yl)
            if relu:
                z = torch.relu(z)
            z.backward(dout)
            rec['bn%d.dout' % l] = dout
            rec['bn%d.din' % l] = yl.grad.float().double()
            for k, p in zip(('bn%d.weight' % l, 'bn%d.bias' % l), ps):
                grads[k] = p.grad.clone()
            dout = rec['bn%d.din' % l]
        # Linear backward: the incoming gradient is cast to bf16, dx is stored bf16, dW / db are fp32 sums
        dyb = _rb(dout)
        rec['lin%d.dout' % l] = dout
        wb = _rb(P['lin%d.weight' % l])
        rec['lin%d.din' % l] = _rb(dyb @ wb.t())
        grads['lin%d.weight' % l] = _rb(rec['lin%d.in' % l]).t() @ dyb
        if 'lin%d.bias' % l in P:
            grads['lin%d.bias' % l] = dyb.sum(0)
        dout = rec['lin%d.din' % l]
    return P, rec, grads


@pytest.mark.parametrize('head', ['mocov3_projector', 'mocov3_predictor', 'simsiam_projector', 'simsiam_predictor'])
def test_bn_mlp_heads_teacher_forced_bf16(head):
    """The BatchNorm-1D MLP heads of MoCo-v3 and SimSiam under bf16 compute, op by op against the fp64 evaluation of
    the same storage contract (every op fed the oracle's input / output gradient): Linear outputs and BatchNorm
    outputs / input gradients / d-gamma / d-beta are fp32 results (relative-to-max bound), a Linear's input gradient is
    a bf16 tensor (<= 2 ulp max, 1/2 ulp mean), weight gradients fp32.  With the teacher-forced trunk layers
    (test_r50_layers_teacher_forced_bf16, test_vit_block_teacher_forced_bf16) and the fp32 criterion kernels below,
    every kernel of the MoCo-v3 / SimSiam bf16 step has an element-wise bound of its own; the whole-step checks of
    tests/test_mocov3_gpu.py / test_simsiam_gpu.py (direction and size of gradients) are smoke checks on top."""
    dims, N, last_affine, bias_last = {
        'mocov3_projector': ([768, 4096, 4096, 256], 64, False, False),
        'mocov3_predictor': ([256, 4096, 256], 64, False, False),
        'simsiam_projector': ([2048, 2048, 2048, 2048], 64, False, False),
        'simsiam_predictor': ([2048, 512, 2048], 64, True, True),
    }[head]
    P, rec, grads = _head_oracle(dims, N, seed=23, last_affine=last_affine, bias_last=bias_last)
    hip_config.set_device('gpu')
    hip_config.set_compute_dtype(torch.bfloat16)
    n_lin = len(dims) - 1
    mods, names = [], []
    for l in range(n_lin):
        lin = nn.Linear(dims[l], dims[l + 1], bias_attr=None if ('lin%d.bias' % l) in P else False)
        mods.append(lin)
        names.append('lin%d' % l)
        if ('bn%d.in' % l) in rec:
            affine = ('bn%d.weight' % l) in P
            mods.append(nn.BatchNorm1D(dims[l + 1]) if affine else
                        nn.BatchNorm1D(dims[l + 1], weight_attr=False, bias_attr=False))
            names.append('bn%d' % l)
    seq = torch.nn.Sequential(*mods)
    arena = nn.EncoderArena(seq, trainable=True)
    with torch.no_grad():
        for m, nm in zip(mods, names):
            for pn, p in m.named_parameters():
                p.copy_(P['%s.%s' % (nm, pn)].to(DEV))
    arena.refresh()
    rep = Report('bn_mlp_head_%s' % head)
    for m, nm in zip(mods, names):
        arena.clear_grad()
        x = rec[nm + '.in'].float().to(DEV).requires_grad_(True)
        if nm.startswith('lin'):
            out = m(nn.to_compute(x, torch.bfloat16), out_f32=True)
            rep.f32(nm + ' fwd (fp32 rows)', out, rec[nm + '.out'], 2e-6)
        else:
            l = int(nm[2:])
            out = m(x, relu=l < n_lin - 1)
            rep.f32(nm + ' fwd (fp32 rows)', out, rec[nm + '.out'], 5e-6)
        out.backward(rec[nm + '.dout'].float().to(DEV))
        torch.cuda.synchronize()
        if nm.startswith('lin'):
            rep.bf16(nm + ' d-input', x.grad, rec[nm + '.din'])
            for pn, p in m.named_parameters():
                rep.f32('%s d %s' % (nm, pn), p.grad, grads['%s.%s' % (nm, pn)], 2e-5)
        else:
            # a pre-activation that is zero to within rounding may land on either side of the ReLU (both are valid
            # evaluations); the gradient is discontinuous there and, through d-gamma / d-beta, moves its whole channel:
            # such channels are left out (at most 0.5 % of them may be)
            keep = ~(((out.detach().cpu() > 0) != (rec[nm + '.out'] > 0)).any(0))
            assert float((~keep).float().mean()) <= 5e-3, 'too many ReLU-boundary channels'
            rep.lines.append('%s: %d of %d channels left out (ReLU boundary)' % (nm, int((~keep).sum()), keep.numel()))
            rep.f32(nm + ' d-input', x.grad.cpu()[:, keep], rec[nm + '.din'][:, keep], 2e-5)
            for pn, p in m.named_parameters():
                rep.f32('%s d %s' % (nm, pn), p.grad.cpu()[keep], grads['%s.%s' % (nm, pn)][keep], 2e-5)
    rep.finish()


def test_contrastive_criteria_fp32_kernels_against_float64():
    """What follows the heads in the two v2 methods, all fp32 kernels: MoCo-v3's l2-normalise -> q . k_all^T / T
    (exact-fp32 MFMA GEMM over the keys of every rank, here 4 ranks' worth) -> row cross-entropy with offset labels
    (passl/models/mocov3.py:170-198), and SimSiam's negative cosine similarity (passl/models/simsiam.py:69,93),
    forward and backward against float64."""
    from passl_amd.models.mocov3 import _KeyLogitsFn
    from passl_amd.modeling.heads.clip_head import _RowCEFn
    from passl_amd.loss.simsiam import neg_cosine_similarity
    gen = torch.Generator().manual_seed(31)
    N, D, W, T = 64, 256, 4, 0.2
    q = torch.randn(N, D, generator=gen).requires_grad_(True)
    k_all = torch.nn.functional.normalize(torch.randn(W * N, D, generator=gen), dim=1)
    rank = 2
    labels = torch.arange(N) + N * rank
    qn = torch.nn.functional.normalize(q.double(), dim=1)
    logits = qn @ k_all.double().t() / T
    loss = torch.nn.functional.cross_entropy(logits, labels) * (2 * T)
    loss.backward()
    qd = q.detach().to(DEV).requires_grad_(True)
    alpha = torch.full((1,), 1.0 / T, device=DEV)
    lg = _KeyLogitsFn.apply(qd, k_all.to(DEV), alpha)
    ld = _RowCEFn.apply(lg, labels.to(DEV)) * (2 * T)
    ld.backward()
    rep = Report('contrastive_criteria_fp32')
    rep.f32('mocov3 logits', lg, logits.detach(), 2e-6)
    rep.f32('mocov3 loss', ld.reshape(()), loss.detach().reshape(()), 2e-6)
    rep.f32('mocov3 d q', qd.grad, q.grad, 2e-5)
    p = (torch.randn(N, 2048, generator=gen) * 0.7).requires_grad_(True)
    z = torch.randn(N, 2048, generator=gen)
    ref = -torch.nn.functional.cosine_similarity(p.double(), z.double(), dim=1).mean()
    ref.backward()
    pd = p.detach().to(DEV).requires_grad_(True)
    got = neg_cosine_similarity(pd, z.to(DEV))
    got.backward()
    rep.f32('simsiam -cos', got.reshape(()), ref.detach().reshape(()), 2e-6)
    rep.f32('simsiam d p', pd.grad, p.grad, 2e-5)
    rep.finish()
