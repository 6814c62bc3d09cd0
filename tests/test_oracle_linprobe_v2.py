"""CPU tests that pin the v2 linear-probe oracle (oracle/linprobe_v2.py) against golden vectors produced by running
the reference's own sources end to end — SimSiamLinearProbe / MoCoV3LinearProbe, CombinedLoss, TopkAcc, MomentumLARC /
Momentum, TimmCosine and the classification train / evaluation loops (tests/golden/make_golden_linprobe_v2.py) — plus
known answers of the update rules and the schedule."""
import math

import numpy as np
import pytest
import torch

from oracle import linprobe_v2 as L
from tests import linprobe_v2_util as U


@pytest.mark.parametrize('name', sorted(U.CASES))
def test_oracle_matches_the_reference_run(name):
    z, N, S, classes, steps = U.load(name)
    o = U.make_oracle(name, classes)
    train, ev = U.batches(N, S, classes, steps)
    for s, (x, y) in enumerate(train):
        pre = 's%d_' % s
        w0 = {n: o.st[n].detach().clone() for n in L.HEAD[o.kind]}
        out = o.train_step(x, y)
        amp = 4.0 ** s
        assert abs(out['lr'] - float(z[pre + 'lr'])) < 1e-12, (s, out['lr'], float(z[pre + 'lr']))
        assert abs(float(out['loss']) - float(z[pre + 'loss'])) < 2e-5 * amp, (s, float(out['loss']), float(z[pre + 'loss']))
        assert abs(float(out['top1']) - float(z[pre + 'top1'])) < 1e-9 and abs(float(out['top5']) - float(z[pre + 'top5'])) < 1e-9
        np.testing.assert_allclose(out['scores'][:, :8].numpy(), z[pre + 'scores_head'], atol=5e-5 * amp)
        for n in L.HEAD[o.kind]:
            g_ref, d_ref = z[pre + 'grad/' + n], z[pre + 'delta/' + n]
            np.testing.assert_allclose(out['grads'][n].numpy(), g_ref, atol=2e-5 * amp * max(np.abs(g_ref).max(), 1e-6))
            d = (o.st[n].detach() - w0[n]).numpy()
            np.testing.assert_allclose(d, d_ref, atol=1e-4 * amp * max(np.abs(d_ref).max(), 1e-9))
            assert abs(o.st[n].double().norm().item() - float(z[pre + 'pnorm/' + n])) < 1e-4 * amp * max(1.0, float(z[pre + 'pnorm/' + n]))
    res = o.evaluate(ev)
    for k in ('CELoss', 'loss', 'top1', 'top5', 'metric'):
        tol = 1e-9 if k.startswith('top') or k == 'metric' else 2e-3
        assert abs(res[k] - float(z['eval_' + k])) < tol + 1e-3 * abs(float(z['eval_' + k])) * (0 if tol < 1e-6 else 1), (k, res[k], float(z['eval_' + k]))


def test_timm_cosine_is_read_at_the_current_last_epoch():
    """optimizer.py:117-120: the optimizer evaluates get_lr() itself.  With the default last_epoch = -1 the first step
    runs at warmup_start_lr when there is a warm-up (mocov3 pre-training yaml); without a warm-up the formula divides by
    warmup_steps = 0 (-1 < 0 takes the warm-up branch), which is why the simsiam / linear-probe yamls say
    last_epoch: 0 and start at the base rate."""
    f = U.lr_fn(dict(learning_rate=0.0024, decay_unit='step', epochs=30, step_each_epoch=10, warmup_epoch=4,
                     warmup_prefix=True))
    assert f(0) == 0.0 and abs(f(1) - 0.0024 / 40) < 1e-18 and abs(f(40) - 0.0024) < 1e-18
    assert abs(f(170) - 0.5 * 0.0024 * (1 + math.cos(math.pi * 130 / 260))) < 1e-18
    g = U.lr_fn(dict(learning_rate=1.0, decay_unit='step', epochs=2, step_each_epoch=5))
    with pytest.raises(ZeroDivisionError):
        g(0)
    assert abs(g(1) - 0.5 * (1 + math.cos(math.pi / 10))) < 1e-15
    h = U.lr_fn(dict(learning_rate=1.6, decay_unit='epoch', epochs=90, step_each_epoch=10, last_epoch=0))
    assert h(0) == 1.6 and abs(h(1) - 0.8 * (1 + math.cos(math.pi / 90))) < 1e-15


def test_larc_known_answers():
    """momentum_larc.py:88-105 on a 2-parameter toy: the trust ratio scales (g + wd p), the clip bounds it by 1 / lr,
    a zero-norm parameter takes the raw gradient without decay."""
    o = L.LinearProbeOracle.__new__(L.LinearProbeOracle)
    o.optimizer, o.lr_value, o.mu, o.wd, o.tc, o.clip, o.eps = 'MomentumLARC', 0.5, 0.9, 0.1, 0.02, False, 0.0
    o.exp_avg, o.step_count = {}, 0
    o.st = {'w': torch.tensor([3.0, 4.0]), 'b': torch.zeros(2)}
    o.update({'w': torch.tensor([0.6, 0.8]), 'b': torch.tensor([1.0, -1.0])})
    a = 0.02 * 5.0 / (1.0 + 5.0 * 0.1)                      # |p| = 5, |g| = 1
    v = a * (torch.tensor([0.6, 0.8]) + 0.1 * torch.tensor([3.0, 4.0]))
    assert torch.allclose(o.st['w'], torch.tensor([3.0, 4.0]) - 0.5 * v, atol=1e-7)
    assert torch.allclose(o.st['b'], -0.5 * torch.tensor([1.0, -1.0]))          # |p| = 0: raw gradient
    o.clip, o.exp_avg, o.st = True, {}, {'w': torch.tensor([3.0, 4.0])}
    o.lr_value = 0.01                                        # a / lr = 6.67 -> clipped to 1
    o.update({'w': torch.tensor([0.6, 0.8])})
    assert torch.allclose(o.st['w'], torch.tensor([3.0, 4.0]) - 0.01 * torch.tensor([0.9, 1.2]), atol=1e-7)
