"""SimSiam (tasks/ssl/simsiam) on a real MI355X: the criterion kernel against torch fp64, whole training
steps of passl.models.simsiam against the torch-CPU restatement run live (every gradient, parameter and running
statistic) and against the golden vectors produced by the reference's own sources, reproducibility, and the v2 Engine
driving it from a yaml with the task's two parameter groups."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import simsiam_util as U                       # noqa: E402
from oracle import simsiam as S                # noqa: E402
from passl_amd.hip import ops                  # noqa: E402

DEV = 'cuda'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def relmax(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


def rel_l2(a, b):
    a, b = a.double().cpu().reshape(-1), b.double().cpu().reshape(-1)
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def test_cosine_loss_kernel_vs_torch():
    gen = torch.Generator().manual_seed(4)
    for N, D in ((8, 2048), (37, 130), (256, 2048)):
        a = torch.randn(N, D, generator=gen, dtype=torch.float64).requires_grad_(True)
        b = torch.randn(N, D, generator=gen, dtype=torch.float64)
        a.data[1] *= 1e-6                                   # |a||b| below eps: the clamped branch
        b.data[1] *= 1e-6
        ref = -S.cosine(a, b).mean()
        ref.backward()
        ad, bd = a.detach().float().to(DEV), b.float().to(DEV)
        loss, stats = ops.cosine_loss_fwd(ad, bd)
        assert abs(float(loss) - float(ref)) < 2e-6
        g = torch.tensor([0.5], device=DEV)
        da = ops.cosine_loss_bwd(ad, bd, stats, g)
        assert relmax(da, 0.5 * a.grad) < 2e-5
        loss2, _ = ops.cosine_loss_fwd(ad, bd)
        assert torch.equal(loss, loss2)                     # fixed-order mean


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


def test_step_matches_oracle_elementwise_fp32():
    """One step from the oracle's state (every residual branch switched on): loss, EVERY gradient tensor, every
    parameter after the two-group momentum update, every running statistic (advanced twice: one BatchNorm batch per
    view).  Truth = the restatement in fp64.  SimSiam at random init is badly conditioned (cosines ~ 0, BatchNorm
    over a handful of rows in the projector / predictor and over N x 2 x 2 positions in layer4): a CPU fp32
    evaluation of the SAME restatement already deviates from fp64 by up to a few per cent in single gradient tensors
    (the reference's own fp32 run does too: tests/golden/simsiam_r50_small.npz holds both).  The product's fp32-compute
    step is therefore held to 6x the deviation of the fp32 restatement, tensor by tensor (floor 2e-4; observed: 1-3.4x)."""
    N, size = 64, 64
    o64 = S.SimSiamOracle(seed=0, zero_init_residual=False, dtype=torch.float64, **U.SOLVER)
    o32 = S.SimSiamOracle(seed=0, zero_init_residual=False, **U.SOLVER)
    model, opt = U.build_product(torch.float32)
    U.load_oracle_state(model, o32)
    model.train()
    gen = torch.Generator().manual_seed(5)
    x1 = torch.randn(N, 3, size, size, generator=gen)
    x2 = torch.randn(N, 3, size, size, generator=gen)
    loss = U.product_step(model, opt, x1.to(DEV), x2.to(DEV))
    ref = o64.train_step(x1.double(), x2.double())
    r32 = o32.train_step(x1, x2)
    rep = Report('simsiam_elementwise_float32')
    rep.check('loss', abs(float(loss.detach()) - float(ref['loss'])), max(2e-6, 6 * abs(float(r32['loss']) - float(ref['loss']))))
    ps = dict(model.named_parameters())
    assert set(n for n, p in ps.items() if p.requires_grad) == set(ref['grads'])
    assert ps['encoder.fc.6.bias'].grad is None
    for n, g in ref['grads'].items():
        rep.check('grad-l2/' + n, rel_l2(ps[n].grad, g), max(2e-4, 6 * rel_l2(r32['grads'][n], g)))
    sd = model.state_dict()
    for k, v in o64.st.items():
        if S.is_stat(k):
            def serr(t):
                return float(((t.cpu().double() - v.double()).abs() / v.double().abs().clamp_min(0.1)).max())
            rep.check('stat/' + k, serr(sd[k]), max(2e-5, 6 * serr(o32.st[k])))
        else:
            rep.check('param/' + k, rel_l2(sd[k], v), max(2e-6, 6 * rel_l2(o32.st[k], v)))
    rep.finish()


def _run_against_golden(name, dtype):
    """Free-running steps against the fp64 trajectory of the restatement stored in the golden file, each quantity held
    to 3x the deviation of the REFERENCE's own fp32 run (same file) from that trajectory — plus a floor per dtype:
    the step is chaotic at random init (see test_step_matches_oracle_elementwise_fp32), the reference's fp32 run is
    the yardstick for what a correct reduced-precision evaluation looks like."""
    z = np.load(os.path.join(GOLDEN, name + '.npz'))
    N, size, steps, zero = [int(v) for v in z['meta']]
    oracle0 = S.SimSiamOracle(seed=0, zero_init_residual=bool(zero), **U.SOLVER)
    model, opt = U.build_product(dtype)
    U.load_oracle_state(model, oracle0)
    model.train()
    bf = dtype == torch.bfloat16
    gen = torch.Generator().manual_seed(777)
    watch = [k[len('s0_gradnorm/'):] for k in z.files if k.startswith('s0_gradnorm/')]
    stats = [k[len('s0_stat/'):] for k in z.files if k.startswith('s0_stat/')]
    rep = Report('%s_%s' % (name, str(dtype).split('.')[-1]))

    def bound(key32, key64, floor):
        return max(floor, 3.0 * float(np.max(np.abs(np.asarray(z[key32], dtype=np.float64) - np.asarray(z[key64], dtype=np.float64)))))
    for s in range(steps):
        x1 = torch.randn(N, 3, size, size, generator=gen)
        x2 = torch.randn(N, 3, size, size, generator=gen)
        loss = U.product_step(model, opt, x1.to(DEV), x2.to(DEV))
        pre, p64 = 's%d_' % s, 's%d_f64_' % s
        rep.check(pre + 'loss', abs(float(loss.detach()) - float(z[p64 + 'loss'])),
                  bound(pre + 'loss', p64 + 'loss', (5e-3 if bf else 5e-6) * 4 ** s))
        ps = dict(model.named_parameters())
        sd = model.state_dict()
        for n in watch:
            ref = float(z[p64 + 'gradnorm/' + n])
            g = ps[n].grad.double().norm().item()
            if ref < 1e-12:
                rep.check(pre + 'gradnorm/' + n + ' (zero)', g, 1e-9)
                continue
            # bf16 floor: 0.25 of the norm — except the predictor's last bias, whose gradient is the batch SUM of
            # per-sample cosine gradients that all but cancel: two equally valid fp32 summation orders of the SAME
            # first-stage BatchNorm statistics (128 consecutive rows per partial in the ring kernel, four 4 x 8 patches
            # in csrc/conv3x3_wave.hip; the convolution outputs are bit-identical) move it from 0.217 to 0.272 at
            # step 0 and from 0.279 to 0.157 at step 2 (profiles/r06_simsiam_bf16_statistics_order.txt); every other
            # watched tensor stays below 0.11 under either order
            floor = (0.4 if n == 'predictor.3.bias' else 0.25) if bf else 4e-3
            rep.check(pre + 'gradnorm/' + n, abs(g - ref) / ref,
                      bound(pre + 'gradnorm/' + n, p64 + 'gradnorm/' + n, 0.0) / ref + floor * 2 ** s)
            rep.check(pre + 'pnorm/' + n, abs(sd[n].double().norm().item() - float(z[p64 + 'pnorm/' + n])),
                      bound(pre + 'pnorm/' + n, p64 + 'pnorm/' + n, (2e-4 if bf else 2e-6) * 4 ** s))
        for n in stats:
            err = np.abs(sd[n][:8].double().cpu().numpy() - z[p64 + 'stat/' + n]).max()
            rep.check(pre + 'stat/' + n, float(err), bound(pre + 'stat/' + n, p64 + 'stat/' + n, (3e-2 if bf else 1e-4) * 2 ** s))
    rep.finish()


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_golden_small(dtype):
    _run_against_golden('simsiam_r50_small', dtype)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_golden_zero_init(dtype):
    """The reference's own initialisation: every bottleneck starts as an identity (gamma of bn3 = 0)."""
    _run_against_golden('simsiam_r50_zero_init', dtype)


def test_step_is_bit_reproducible():
    oracle = S.SimSiamOracle(seed=0, zero_init_residual=False, **U.SOLVER)
    gen = torch.Generator().manual_seed(5)
    xs = [(torch.randn(8, 3, 64, 64, generator=gen).to(DEV), torch.randn(8, 3, 64, 64, generator=gen).to(DEV))
          for _ in range(3)]
    ends = []
    for _ in range(2):
        model, opt = U.build_product(torch.bfloat16)
        U.load_oracle_state(model, oracle)
        model.train()
        losses = [U.product_step(model, opt, *x).detach().clone() for x in xs]
        ends.append((torch.cat(losses), model.arena_q.flat.clone(), model.arena_p.flat.clone()))
    for a, b in zip(*ends):
        assert torch.equal(a, b)


def test_v2_engine_trains_simsiam_from_yaml(tmp_path):
    """Engine(config).train() on configs/v2/simsiam_resnet50_pt_synthetic.yaml (the Optimizer block of the
    reference's tasks/ssl/simsiam yaml: Momentum, `lr` = TimmCosine per EPOCH inside the Optimizer block,
    param_groups encoder / predictor with the predictor at a fixed 0.1): two epochs of two steps reproduce the
    restatement, and the two groups really run at different rates after the first epoch."""
    from passl.engine.engine import Engine, OptimizerGroup
    from passl_amd.utils.config import get_config
    N = 16
    cfg = get_config(os.path.join(ROOT, 'configs', 'v2', 'simsiam_resnet50_pt_synthetic.yaml'),
                     ['Global.epochs=2', 'Global.output_dir=%s' % tmp_path, 'Global.print_batch_step=1',
                      'DataLoader.Train.dataset.num_samples=%d' % (N * 2), 'DataLoader.Train.dataset.image_size=64',
                      'DataLoader.Train.sampler.batch_size=%d' % N,
                      # small rates: at the yaml's 0.1 the random-init step is chaotic beyond the first update
                      'Optimizer.lr.learning_rate=0.001', 'Optimizer.param_groups.1.lr=0.002'])
    eng = Engine(cfg, mode='train')
    assert type(eng.model).__name__ == 'SimSiamPretain' and isinstance(eng.optimizer, OptimizerGroup)
    assert eng.optimizer.names == ['encoder', 'predictor'] and eng.lr_decay_unit == 'epoch'
    assert type(eng.lr_scheduler).__name__ == 'TimmCosine' and eng.lr_scheduler.T_max == 2
    lrs = []
    oracle = S.SimSiamOracle(seed=3, zero_init_residual=False, lr=lambda step: lrs[step], predictor_lr=0.002,
                             momentum=0.9, weight_decay=1e-4)
    U.load_oracle_state(eng.model, oracle)
    x1, x2 = (t.cpu() for t in eng.train_dataloader.inner._cache[0])
    losses = []
    inner = eng.train_loop.train_one_step

    def spy(batch):
        lrs.append(eng.optimizer.get_lr(0))
        assert eng.optimizer.get_lr(1) == 0.002
        out, ld = inner(batch)
        losses.append(ld['loss'].detach().reshape(()).clone())
        return out, ld
    eng.train_loop.train_one_step = spy
    eng.train()
    assert eng.global_step == 4 and len(losses) == 4
    assert lrs[0] == lrs[1] == 0.001 and abs(lrs[2] - 0.0005) < 1e-12 and lrs[3] == lrs[2]      # cosine over 2 epochs
    ref = [oracle.train_step(x1, x2) for _ in range(4)]
    got = [float(v) for v in losses]
    for s_ in range(4):
        # (fp32 against fp32 at a badly conditioned point: see test_step_matches_oracle_elementwise_fp32)
        assert abs(got[s_] - float(ref[s_]['loss'])) < 2e-4 * 4 ** s_, (s_, got, [float(r['loss']) for r in ref])
