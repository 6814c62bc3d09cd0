"""CPU tests that pin the oracle: (1) against golden vectors produced by running the
reference's own sources (tests/golden/make_golden.py), (2) live against the
reference when /root/reference is present, (3) fp64 / invariant known-answer tests
(SURVEY §8c)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import golden_util as G
from oracle import moco as M
from oracle import resnet50 as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_oracle_against(name, max_steps):
    z, N, hw, K, steps = G.load(name)
    o = M.MoCoOracle(K=K, seed=0, t_max=200 * 5004, **G.oracle_kwargs(name))
    gen = torch.Generator().manual_seed(1234)
    for s in range(min(steps, max_steps)):
        xq, xk = G.views(gen, N, hw)
        ptr0 = o.queue_ptr
        out = o.train_step(xq, xk)
        pre = 's%d_' % s
        assert abs(float(out['loss']) - float(z[pre + 'loss'])) < 1e-5
        assert float(out['acc1']) == float(z[pre + 'acc1'])
        assert float(out['acc5']) == float(z[pre + 'acc5'])
        assert o.queue_ptr == int(z[pre + 'queue_ptr'])
        np.testing.assert_allclose(out['logits'][:, :8].numpy(), z[pre + 'logits_head'],
                                   rtol=0, atol=1e-5)
        np.testing.assert_allclose(o.queue[:, ptr0:ptr0 + N].numpy(), z[pre + 'queue_new'],
                                   rtol=0, atol=1e-6)
        for n in G.watch(name):
            g = out['grads'][n].double().norm().item()
            assert abs(g - float(z[pre + 'gradnorm/' + n])) <= 1e-4 * max(g, 1e-6), n
            assert abs(o.q[n].double().norm().item() - float(z[pre + 'qnorm/' + n])) < 1e-4
            assert abs(o.k[n].double().norm().item() - float(z[pre + 'knorm/' + n])) < 1e-4
        for n in G.WATCH_STATS:
            np.testing.assert_allclose(o.q[n][:8].numpy(), z[pre + 'qstat/' + n], atol=1e-5)
            np.testing.assert_allclose(o.k[n][:8].numpy(), z[pre + 'kstat/' + n], atol=1e-5)


def test_oracle_matches_golden_small():
    _run_oracle_against('moco_v2_r50_small', 3)


def test_oracle_matches_golden_v1_small():
    """configs/moco/moco_v1_r50.yaml: LinearNeck projector, T = 0.07, MultiStepDecay."""
    _run_oracle_against('moco_v1_r50_small', 3)
    o = M.MoCoOracle(K=256, seed=0, **G.V1)
    assert o.lr() == 0.03
    o.step_count = 120 * 5004
    assert abs(o.lr() - 0.003) < 1e-12
    o.step_count = 160 * 5004
    assert abs(o.lr() - 0.0003) < 1e-12


def test_oracle_matches_golden_cfg1_first_step():
    _run_oracle_against('moco_v2_r50_cfg1', 1)


@pytest.mark.skipif(not os.path.isdir('/root/reference/passl_v110'),
                    reason='reference tree not present (GPU box)')
def test_oracle_matches_reference_sources_live():
    """Executes the reference's moco.py under the paddle shim in a subprocess
    (the shim monkey-patches torch.Tensor) and compares with the oracle."""
    code = r'''
import torch
from oracle import ref_runner
from oracle.moco import MoCoOracle
model = ref_runner.build_reference_moco(K=256)
o = MoCoOracle(K=256)
ref_runner.load_oracle_state(model, o)
model.train()
g = torch.Generator().manual_seed(7)
for step in range(2):
    xq = torch.randn(4, 3, 96, 96, generator=g); xk = torch.randn(4, 3, 96, 96, generator=g)
    for p in model.parameters(): p.grad = None
    res = model(xq, xk, mode='train', total_iters=2, current_iter=step + 1, mixup_fn=None)
    res['loss'].backward()
    qs = dict(model.encoder_q.named_parameters())
    o_q_before = {n: t.detach().clone() for n, t in o.q.items()}
    out = o.train_step(xq, xk)
    assert abs(float(res['loss']) - float(out['loss'])) < 1e-6, (float(res['loss']), float(out['loss']))
    assert float(res['acc1']) == float(out['acc1']) and float(res['acc5']) == float(out['acc5'])
    for n, gr in out['grads'].items():
        assert (qs[n].grad - gr).abs().max().item() < 1e-6, n
    assert (model.queue - o.queue).abs().max().item() == 0.0
    assert int(model.queue_ptr[0]) == o.queue_ptr
    ks = model.encoder_k.state_dict()
    for n in o.k:
        assert (ks[n] - o.k[n]).abs().max().item() < 1e-7, n
    # mirror the oracle's optimizer step into the reference model for the next iteration
    with torch.no_grad():
        for n, p in model.encoder_q.state_dict().items():
            p.copy_(o.q[n])
print('REFERENCE_MATCH_OK')
'''
    r = subprocess.run([sys.executable, '-c', code], cwd=ROOT, capture_output=True, text=True,
                       timeout=600)
    assert 'REFERENCE_MATCH_OK' in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_head_fp32_vs_fp64_and_invariants():
    g = torch.Generator().manual_seed(3)
    N, D, K, T = 16, 128, 4096, 0.2
    q = M.l2_normalize(torch.randn(N, D, generator=g), 1)
    k = M.l2_normalize(torch.randn(N, D, generator=g), 1)
    queue = M.l2_normalize(torch.randn(D, K, generator=g), 0)
    pos = (q * k).sum(1, keepdim=True)
    neg = q @ queue
    loss, a1, a5, logits = M.contrastive_head(pos, neg, T)
    l64, a1_64, a5_64, lg64 = M.contrastive_head_f64(pos.numpy(), neg.numpy(), T)
    assert abs(float(loss) - l64) < 1e-5
    assert float(a1) == a1_64 and float(a5) == a5_64
    assert np.abs(logits.numpy() * T).max() <= 1.0 + 1e-5          # |cos| <= 1
    np.testing.assert_allclose(logits[:, 0].numpy() * T, (q * k).sum(1).numpy(), atol=1e-6)
    # loss at "no information" ~ ln(K+1) within O(1/T)
    assert abs(l64 - np.log(K + 1)) < 2.0 / T


def test_queue_and_ptr_invariants():
    o = M.MoCoOracle(K=64, width_div=16, t_max=100)
    g = torch.Generator().manual_seed(11)
    for s in range(3):
        xq = torch.randn(4, 3, 32, 32, generator=g)
        xk = torch.randn(4, 3, 32, 32, generator=g)
        out = o.train_step(xq, xk)
        assert o.queue_ptr == ((s + 1) * 4) % 64
        np.testing.assert_allclose(o.queue.norm(dim=0).numpy(), 1.0, atol=1e-5)
        # key encoder never receives gradients; q encoder does
        assert all(not t.requires_grad for t in o.k.values())
        assert set(out['grads']) == set(R.trainable_keys(o.q))


def test_momentum_rule_first_and_second_step():
    """optimizer: p1 = p0 - lr*(g + wd*p0); v2 = mu*v1 + (g2 + wd*p1)."""
    o = M.MoCoOracle(K=64, width_div=16, t_max=10)
    n = '1.mlp.2.bias'
    p0 = o.q[n].clone()
    g1 = torch.full_like(p0, 0.5)
    lr0 = o.lr()
    assert lr0 == 0.015
    o.apply_momentum({n: g1})
    v1 = g1 + o.wd * p0
    np.testing.assert_allclose(o.q[n].numpy(), (p0 - lr0 * v1).numpy(), rtol=1e-6)
    lr1 = o.lr()
    assert abs(lr1 - 0.015 * 0.5 * (1 + np.cos(np.pi * 1 / 10))) < 1e-12
    p1 = o.q[n].clone()
    o.apply_momentum({n: g1})
    v2 = o.mu * v1 + (g1 + o.wd * p1)
    np.testing.assert_allclose(o.q[n].numpy(), (p1 - lr1 * v2).numpy(), rtol=1e-6)


def test_bn_paddle_semantics_restated():
    """Training BN: biased batch var for normalisation AND for the running update,
    momentum 0.9; frozen BN uses running stats and leaves them untouched."""
    g = torch.Generator().manual_seed(5)
    x = torch.randn(6, 4, 5, 5, generator=g) * 3 + 1
    st = {'p.weight': torch.rand(4, generator=g) + .5, 'p.bias': torch.randn(4, generator=g),
          'p._mean': torch.randn(4, generator=g), 'p._variance': torch.rand(4, generator=g) + .5}
    new = {}
    y = R.batch_norm(x, st, 'p', use_global_stats=False, new_stats=new)
    x64 = x.double().numpy()
    mu = x64.mean(axis=(0, 2, 3)); var = x64.var(axis=(0, 2, 3))
    ref = (x64 - mu[None, :, None, None]) / np.sqrt(var + 1e-5)[None, :, None, None] \
        * st['p.weight'].double().numpy()[None, :, None, None] + st['p.bias'].double().numpy()[None, :, None, None]
    np.testing.assert_allclose(y.numpy(), ref, atol=1e-5)
    np.testing.assert_allclose(new['p._mean'].numpy(), 0.9 * st['p._mean'].numpy() + 0.1 * mu, atol=1e-6)
    np.testing.assert_allclose(new['p._variance'].numpy(), 0.9 * st['p._variance'].numpy() + 0.1 * var, atol=1e-6)
    y2 = R.batch_norm(x, st, 'p', use_global_stats=True)
    ref2 = (x64 - st['p._mean'].double().numpy()[None, :, None, None]) / np.sqrt(st['p._variance'].double().numpy() + 1e-5)[None, :, None, None] \
        * st['p.weight'].double().numpy()[None, :, None, None] + st['p.bias'].double().numpy()[None, :, None, None]
    np.testing.assert_allclose(y2.numpy(), ref2, atol=1e-5)


def test_parameter_counts_match_survey():
    st = R.init_encoder_state(torch.Generator().manual_seed(0))
    tk = R.trainable_keys(st)
    assert sum(st[k].numel() for k in tk) == 27966656
    assert sum(st[k].numel() for k in st if k.endswith('_mean') or k.endswith('_variance')) == 2 * 26560
    assert len([k for k in st if k.endswith('.weight') and st[k].dim() == 4]) == 53
