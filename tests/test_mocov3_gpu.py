"""MoCo-v3 on a real MI355X: whole training steps of passl.models.mocov3 on the HIP kernels against
(a) the torch-CPU restatement of the reference run live on the same inputs — every gradient and every updated
parameter, element by element, from the SAME state (teacher forced), on both branches of the momentum update —
and (b) the golden vectors produced by the reference's own sources (tests/golden/mocov3_*.npz), plus the v2 Engine
driving the model from a yaml."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import mocov3_util as U                        # noqa: E402
from oracle import mocov3 as O                 # noqa: E402

DEV = 'cuda'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')
WATCH = ['base_encoder.cls_token', 'base_encoder.blocks.0.attn.qkv.weight', 'base_encoder.blocks.1.mlp.fc2.bias',
         'base_encoder.blocks.1.norm2.weight', 'base_encoder.head.0.weight', 'base_encoder.head.4.bias',
         'base_encoder.head.6.weight', 'predictor.0.weight', 'predictor.1.weight', 'predictor.3.weight']
WATCH_MOM = ['base_encoder.patch_embed.proj.weight', 'base_encoder.blocks.0.attn.qkv.weight',
             'base_encoder.head.6.weight', 'predictor.3.weight', 'base_encoder.head.7._mean',
             'predictor.1._variance']
# analytically zero gradients (rounding noise that AdamW turns into +-lr steps): the shift in front of a
# BatchNorm-ed bias-free Linear, and the key bias (softmax is invariant to a per-query constant)
NOISE_ONLY = ('base_encoder.norm.bias',)


def relmax(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


class Report(object):
    def __init__(self, name):
        self.name, self.lines, self.bad = name, [], []

    def check(self, what, err, bound):
        line = '%-64s err %.3e  bound %.1e' % (what, err, bound)
        self.lines.append(line)
        if not err <= bound:
            self.bad.append(line)

    def finish(self):
        print('\n'.join(self.lines))
        try:
            os.makedirs('gpurun_out', exist_ok=True)
            with open('gpurun_out/parity_%s.txt' % self.name, 'w') as f:
                f.write('\n'.join(self.lines) + '\n\nVIOLATIONS (%d)\n' % len(self.bad) + '\n'.join(self.bad) + '\n')
        except OSError:
            pass
        assert not self.bad, 'parity violations:\n' + '\n'.join(self.bad)


# fp32 compute pins the algorithm: every gradient to ~1e-5.  bf16 compute is bounded by the conditioning of the model,
# not by the kernels (each of which is held to a few bf16 ulp on bf16 inputs in test_mae_gpu.py / test_ops_gpu.py):
# five BatchNorm layers over a handful of nearly equal rows amplify the 2^-9 input rounding to ~2.5 % of a standard
# deviation in q and k, and the temperature 0.2 turns that into 10-20 % in d loss / d logits (scratch/
# mocov3_bf16_probe.py, profiles/r03_mocov3_bf16_conditioning.txt: same numbers from N = 8 to N = 128).  The bf16
# checks are therefore direction (cosine) and size (relative L2) of every gradient tensor.
TOL = {torch.float32: dict(loss=2e-5, grad=2e-4, param=5e-6, mom=2e-5, stat=2e-4),
       torch.bfloat16: dict(loss=3e-2, grad=0.40, cos=0.92, param=2e-4, mom=2e-3, stat=5e-2)}


def rel_l2(a, b):
    a, b = a.double().cpu().reshape(-1), b.double().cpu().reshape(-1)
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def cosine(a, b):
    a, b = a.double().cpu().reshape(-1), b.double().cpu().reshape(-1)
    return float((a @ b) / (a.norm() * b.norm()).clamp_min(1e-30))


def stat_err(got, ref):
    """BatchNorm running statistics: the mean behind a BatchNorm-ed (zero-mean) input is itself ~0, so the error is
    taken relative to max(|ref|, 0.1) (the statistics start at mean 0 / variance 1 and move by 10 % per step)."""
    got, ref = got.double().cpu(), ref.double().cpu()
    return float(((got - ref).abs() / ref.abs().clamp_min(0.1)).max())


def _is_kbias_noise(name):
    return name.endswith('attn.qkv.bias')


@pytest.mark.parametrize('start_steps', [0, 3])
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_step_matches_oracle_elementwise(dtype, start_steps):
    """One step from the oracle's state: loss, EVERY gradient, every parameter after AdamW, the whole momentum
    encoder after its update (start_steps 0: the copy branch; 3: the cosine-momentum average, from a momentum
    encoder that differs from the base encoder) and its BatchNorm statistics after its own two forward passes."""
    cfg = O.SMALL
    bf = dtype == torch.bfloat16
    N = 32 if bf else 8
    tol = TOL[dtype]
    oracle = O.MoCoV3Oracle(cfg, seed=0, max_steps=10, **U.SOLVER)
    gen = torch.Generator().manual_seed(5)
    if start_steps:
        oracle.steps = start_steps
        for k in oracle.mom:
            if not O.is_buffer(k) and k not in O.FROZEN:
                oracle.mom[k] = oracle.mom[k] + 0.02 * torch.randn(oracle.mom[k].shape, generator=gen)
            elif k.endswith('._variance'):
                oracle.mom[k] = oracle.mom[k] + 0.1
    model, opt = U.build_product(cfg, dtype, max_steps=10)
    U.load_oracle_state(model, oracle)
    model.train()
    x1 = torch.randn(N, 3, 64, 64, generator=gen)
    x2 = torch.randn(N, 3, 64, 64, generator=gen)
    loss = U.product_step(model, opt, x1.to(DEV), x2.to(DEV))
    ref = oracle.train_step(x1, x2)
    rep = Report('mocov3_elementwise_%s_steps%d' % (str(dtype).split('.')[-1], start_steps))
    rep.check('loss', abs(float(loss.detach()) - float(ref['loss'])), tol['loss'])
    assert model.momentum_encoder._steps == oracle.steps == start_steps + 1
    assert int(model.momentum_encoder.steps) == oracle.steps
    ps = dict(model.named_parameters())
    assert set(n for n, p in ps.items() if p.requires_grad) == set(ref['grads'])
    gmax = max(float(g.abs().max()) for n, g in ref['grads'].items() if n != 'base_encoder.cls_token')
    for n, g in ref['grads'].items():
        if n in NOISE_ONLY:
            continue
        got = ps[n].grad
        if _is_kbias_noise(n):           # [q | k | v] bias: the k third is analytically zero
            D = g.numel() // 3
            rep.check('grad/' + n + '[k] (zero)', float(got[D:2 * D].abs().max()) / gmax, 2e-5 if not bf else 2e-2)
            got, g = torch.cat([got[:D], got[2 * D:]]), torch.cat([g[:D], g[2 * D:]])
            n = n + '[q,v]'
        if bf:
            rep.check('grad-l2/' + n, rel_l2(got, g), tol['grad'])
            rep.check('grad-cos/' + n, 1.0 - cosine(got, g), 1.0 - tol['cos'])
        else:
            rep.check('grad/' + n, relmax(got, g), tol['grad'])
    sd = model.state_dict()
    lr = U.SOLVER['lr']
    for n in ref['grads']:
        if n in NOISE_ONLY or _is_kbias_noise(n):
            continue
        # AdamW's first step is lr * g / (|g| + eps): elements whose gradient is at rounding level may take the
        # step in either direction; everything else must agree to rounding
        diff = (sd[n].cpu().double() - oracle.st[n].double()).abs()
        small = ref['grads'][n].abs().double() < (1e-6 if dtype == torch.float32 else 2e-2) * gmax
        rep.check('param/' + n, float(diff[~small].max()) if (~small).any() else 0.0,
                  tol['param'] if dtype == torch.float32 else 2.1 * lr)
        assert float(diff.max()) <= 2.1 * lr, n
    for k, v in oracle.mom.items():
        got = sd[U.mom_key(k)]
        rep.check('momentum/' + k, stat_err(got, v) if O.is_buffer(k) else relmax(got, v),
                  tol['stat'] if O.is_buffer(k) else tol['mom'])
    for k, v in oracle.st.items():
        if O.is_buffer(k):
            rep.check('bn-stat/' + k, stat_err(sd[k], v), tol['stat'])
    rep.finish()


def _run_against_golden(name, cfg, N, steps, max_steps, dtype):
    """Free-running steps against the reference's own run (fp32 torch-CPU through the paddle shim) and the fp64
    trajectory of the restatement stored beside it.  Step 0 is exact to the dtype; later steps start from
    parameters that moved by +-lr where a gradient was at rounding level, amplified ~10x per step by the tiny
    BatchNorm batches (the reference run and the fp32 restatement drift apart the same way:
    tests/test_oracle_mocov3.py)."""
    z = np.load(os.path.join(GOLDEN, name + '.npz'))
    oracle0 = O.MoCoV3Oracle(cfg, seed=0, max_steps=max_steps, **U.SOLVER)
    model, opt = U.build_product(cfg, dtype, max_steps=max_steps)
    U.load_oracle_state(model, oracle0)
    model.train()
    gen = torch.Generator().manual_seed(777)
    S = cfg['img_size']
    bf = dtype == torch.bfloat16
    rep = Report('%s_%s' % (name, str(dtype).split('.')[-1]))
    for s in range(steps):
        x1 = torch.randn(N, 3, S, S, generator=gen)
        x2 = torch.randn(N, 3, S, S, generator=gen)
        loss = U.product_step(model, opt, x1.to(DEV), x2.to(DEV))
        amp = 30.0 ** s
        pre = 's%d_' % s
        for tag in ('', 'f64_'):
            rep.check(pre + tag + 'loss', abs(float(loss.detach()) - float(z[pre + tag + 'loss'])),
                      (3e-2 if bf else 2e-5) * amp)
        assert model.momentum_encoder._steps == int(z[pre + 'ema_steps'])
        ps = dict(model.named_parameters())
        sd = model.state_dict()
        for n in WATCH:
            g = ps[n].grad.double().norm().item()
            rep.check(pre + 'gradnorm/' + n, abs(g - float(z[pre + 'f64_gradnorm/' + n])) / float(z[pre + 'f64_gradnorm/' + n]),
                      0.45 if bf else 3e-4 * amp)
            rep.check(pre + 'pnorm/' + n, abs(sd[n].double().norm().item() - float(z[pre + 'f64_pnorm/' + n])),
                      (2e-3 if bf else 5e-5) * amp)
        for n in WATCH_MOM:
            v = sd[U.mom_key(n)].double().norm().item()
            rep.check(pre + 'mom_pnorm/' + n, abs(v - float(z[pre + 'f64_mom_pnorm/' + n])) / max(v, 1.0),
                      (2e-2 if bf else 2e-5) * amp)
    rep.finish()


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_golden_small(dtype):
    _run_against_golden('mocov3_small', O.SMALL, 8, 3, 10, dtype)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_golden_vit_b(dtype):
    """tasks/ssl/mocov3/configs/mocov3_vit_base_patch16_224_pt_in1k_4n32c_dp_fp16o1.yaml's architecture
    (ViT-B/16, 197 tokens, 4096-wide MLPs, dim 256)."""
    _run_against_golden('mocov3_vit_b', O.VIT_B, 4, 2, 100, dtype)


def test_step_is_bit_reproducible():
    """Two runs from the same state on the same inputs end in bit-identical parameters (no atomics on this path's
    reductions that would reorder sums between runs)."""
    cfg, N = O.SMALL, 8
    oracle = O.MoCoV3Oracle(cfg, seed=0, max_steps=10, **U.SOLVER)
    gen = torch.Generator().manual_seed(5)
    xs = [(torch.randn(N, 3, 64, 64, generator=gen).to(DEV), torch.randn(N, 3, 64, 64, generator=gen).to(DEV))
          for _ in range(3)]
    out = []
    for _ in range(2):
        model, opt = U.build_product(cfg, torch.bfloat16, max_steps=10)
        U.load_oracle_state(model, oracle)
        model.train()
        losses = [U.product_step(model, opt, *x).detach().clone() for x in xs]
        out.append((torch.cat(losses), model.arena_q.flat.clone(), model.arena_k.flat.clone()))
    for a, b in zip(*out):
        assert torch.equal(a, b)


def test_v2_engine_trains_mocov3_from_yaml(tmp_path):
    """Engine(config).train() on configs/v2/mocov3_vit_base_pt_synthetic.yaml (the reference yaml's Model /
    LRScheduler / Optimizer / FP16 blocks over the synthetic source): ViT-B/16 MoCo-v3, AdamW, TimmCosine,
    ContrastiveLearningTrainingEpochLoop; three steps reproduce the restatement run on the loader's batch with the
    schedule's learning rates as the reference's optimizer reads them (get_lr() at the current last_epoch,
    passl/optimizer/optimizer.py:117-120: warmup_start_lr = 0 on the first step — AdamW moves nothing —, then
    k * peak / warmup_steps after lr_step(k))."""
    from passl.engine.engine import Engine
    from passl_amd.utils.config import get_config
    from passl_amd.utils.infohub import runtime_info_hub
    N, steps = 4, 3
    cfg = get_config(os.path.join(ROOT, 'configs', 'v2', 'mocov3_vit_base_pt_synthetic.yaml'),
                     ['Global.epochs=2', 'Global.output_dir=%s' % tmp_path, 'Global.print_batch_step=1',
                      'DataLoader.Train.dataset.num_samples=%d' % (N * 10),
                      'DataLoader.Train.sampler.batch_size=%d' % N])
    cfg['Global']['max_train_step'] = steps
    cfg['Global']['compute_dtype'] = 'fp32'
    eng = Engine(cfg, mode='train')
    assert type(eng.model).__name__ == 'MoCoV3Pretrain' and runtime_info_hub.max_steps == 20
    sched = eng.lr_scheduler
    assert type(sched).__name__ == 'TimmCosine' and sched.T_max == 20
    lrs = []
    oracle = O.MoCoV3Oracle(O.VIT_B, seed=3, max_steps=20, lr=lambda step: lrs[step], beta1=0.9, beta2=0.999,
                            eps=1e-8, weight_decay=0.1)
    U.load_oracle_state(eng.model, oracle)
    x1, x2 = (t.cpu() for t in eng.train_dataloader.inner._cache[0])
    losses = []
    inner = eng.train_loop.train_one_step

    def spy(batch):
        lrs.append(eng.optimizer.get_lr())
        out, ld = inner(batch)
        losses.append(ld['loss'].detach().reshape(()).clone())
        return out, ld
    eng.train_loop.train_one_step = spy
    eng.train()
    assert eng.global_step == steps and len(losses) == steps
    warm = sched.warmup_steps
    assert lrs[0] == 0.0 and abs(lrs[1] - 0.0024 / warm) < 1e-15 and abs(lrs[2] - 2 * 0.0024 / warm) < 1e-15
    ref = [oracle.train_step(x1, x2) for _ in range(steps)]
    got = [float(v) for v in losses]
    for s_ in range(steps):
        assert abs(got[s_] - float(ref[s_]['loss'])) < 5e-5 * 10 ** s_, (got, [float(r['loss']) for r in ref])
    assert eng.model.momentum_encoder._steps == steps
