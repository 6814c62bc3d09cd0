"""Shared by the v2 linear-probe tests (CPU oracle and GPU product): the golden cases of
tests/golden/make_golden_linprobe_v2.py and the learning rate the reference's loop applies at each step."""
import os

import numpy as np
import torch

from oracle import linprobe_v2 as L
from oracle.mocov3 import SMALL

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
VIT_SMALL = dict(SMALL)

CASES = {
    'lp_simsiam_r50': dict(kind='simsiam',
                           opt=dict(optimizer='MomentumLARC', momentum=0.9, weight_decay=0.0, trust_coefficient=0.001,
                                    clip=False),
                           sched=dict(learning_rate=1.6, decay_unit='step', epochs=2, step_each_epoch=3, last_epoch=0)),
    'lp_simsiam_r50_clip': dict(kind='simsiam',
                                opt=dict(optimizer='MomentumLARC', momentum=0.9, weight_decay=1e-3,
                                         trust_coefficient=0.02, clip=True),
                                sched=dict(learning_rate=0.05, decay_unit='step', epochs=2, step_each_epoch=3,
                                           last_epoch=0)),
    'lp_mocov3_small': dict(kind='mocov3',
                            opt=dict(optimizer='Momentum', momentum=0.9, weight_decay=0.0),
                            sched=dict(learning_rate=0.5, decay_unit='step', epochs=4, step_each_epoch=2,
                                       warmup_epoch=1, warmup_start_lr=0.01)),
}


def lr_fn(sched):
    """Optimizer step s (0-based) of a run that steps the schedule with ``lr_step(global_step)`` after every step:
    get_lr() at the constructor's last_epoch for s = 0, at last_epoch = s afterwards."""
    sc = dict(sched)
    unit = sc.pop('decay_unit')
    epochs, per = sc.pop('epochs'), sc.pop('step_each_epoch')
    first = sc.pop('last_epoch', -1)
    warm = sc.pop('warmup_epoch', 0)
    T_max = epochs * per if unit == 'step' else epochs
    warm = int(round(warm * per)) if unit == 'step' else warm
    return lambda s: L.timm_cosine(first if s == 0 else s, sc['learning_rate'], T_max, warm,
                                   sc.get('eta_min', 0.0), sc.get('warmup_start_lr', 0.0),
                                   sc.get('warmup_prefix', False))


def load(name):
    z = np.load(os.path.join(GOLDEN, name + '.npz'))
    N, S, classes, steps = [int(v) for v in z['meta']]
    return z, N, S, classes, steps


def batches(N, S, classes, steps):
    """-> (training batches, evaluation batches) as the generator script draws them."""
    gen = torch.Generator().manual_seed(4242)
    train = []
    for _ in range(steps):
        x = torch.randn(N, 3, S, S, generator=gen)
        train.append((x, torch.randint(0, classes, (N,), generator=gen)))
    ev = []
    for n in (N, N // 2):
        x = torch.randn(n, 3, S, S, generator=gen)
        ev.append((x, torch.randint(0, classes, (n,), generator=gen)))
    return train, ev


def make_oracle(name, classes, dtype=torch.float32):
    c = CASES[name]
    return L.LinearProbeOracle(c['kind'], class_num=classes, seed=0, cfg=VIT_SMALL if c['kind'] == 'mocov3' else None,
                               lr=lr_fn(c['sched']), dtype=dtype, **c['opt'])
