"""Helpers shared by the golden-vector tests (CPU oracle and GPU product path)."""
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')

WATCH = ['0.conv1.weight', '0.layer1.0.conv2.weight', '0.layer2.0.downsample.0.weight',
         '0.layer4.2.conv3.weight', '0.layer3.5.bn2.weight', '0.bn1.bias',
         '1.mlp.0.weight', '1.mlp.2.weight', '1.mlp.2.bias']
WATCH_STATS = ['0.bn1._mean', '0.bn1._variance', '0.layer4.2.bn3._mean',
               '0.layer4.2.bn3._variance']


# configs/moco/moco_v1_r50.yaml: LinearNeck, T = 0.07, lr 0.03 MultiStepDecay([120, 160] epochs)
V1 = dict(neck='LinearNeck', T=0.07, lr=0.03, milestones=[120 * 5004, 160 * 5004])


def is_v1(name):
    return name.startswith('moco_v1')


def oracle_kwargs(name):
    return dict(V1) if is_v1(name) else {}


def watch(name):
    if not is_v1(name):
        return WATCH
    return [n for n in WATCH if not n.startswith('1.')] + ['1.fc.weight', '1.fc.bias']


def load(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + '.npz'))
    N, hw, K, steps = [int(v) for v in z['meta']]
    return z, N, hw, K, steps


def views(gen, N, hw):
    """Same draw order as tests/golden/make_golden.py."""
    xq = torch.randn(N, 3, hw, hw, generator=gen)
    xk = torch.randn(N, 3, hw, hw, generator=gen)
    return xq, xk


def rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-12))
