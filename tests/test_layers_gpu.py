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
