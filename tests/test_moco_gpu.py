"""Whole-step parity on a real MI355X: the HIP MoCo-v2 R50 step (through the registries and the
C ABI) against (a) the golden vectors produced by running the reference's own sources
(tests/golden/*.npz) and (b) the CPU oracle run live at a small size.

Tolerances: fp32 compute (exact-fp32 MFMA) must meet BASELINE.json's 1e-3 on loss / logits /
queue; bf16 compute (the benchmark dtype) is compared with the same fp32 goldens at a stated
looser bound because the reference has no bf16 MoCo path."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import golden_util as G                       # noqa: E402
import moco_util as U                         # noqa: E402
from oracle.moco import MoCoOracle            # noqa: E402
from passl_amd.hip import ops                 # noqa: E402

DEV = 'cuda'


def _run_against_golden(name, dtype, steps_cap, tol):
    z, N, hw, K, steps = G.load(name)
    oracle0 = MoCoOracle(K=K, seed=0, t_max=200 * 5004)     # seed-defined initial state
    model, opt, sched = U.build_product(K, dtype)
    U.load_oracle_state(model, oracle0)
    model.train()
    captured = {}
    fused = model.head.fused

    def spy(q, k, queue):
        captured.update(q=q.detach(), k=k.detach(), queue=queue)
        return fused(q, k, queue)
    model.head.fused = spy
    gen = torch.Generator().manual_seed(1234)
    for s in range(min(steps, steps_cap)):
        xq, xk = G.views(gen, N, hw)
        ptr0 = model._ptr
        out = U.product_step(model, opt, sched, xq.to(DEV), xk.to(DEV))
        pre = 's%d_' % s
        loss = float(out['loss'])
        assert abs(loss - float(z[pre + 'loss'])) < tol['loss'], (s, loss, float(z[pre + 'loss']))
        if tol['exact_acc']:
            assert abs(float(out['acc1']) - float(z[pre + 'acc1'])) < 1e-3
            assert abs(float(out['acc5']) - float(z[pre + 'acc5'])) < 1e-3
        # logits (recomputed by the same kernel with materialisation on)
        _o, lse, logits = ops.infonce_fwd(captured['q'], captured['k'], captured['queue'],
                                          model.head.temperature, want_logits=True)
        logits = logits.double().cpu()
        assert np.abs(logits[:, :8].numpy() - z[pre + 'logits_head']).max() < tol['logits']
        assert np.abs(lse.double().cpu().numpy() - z[pre + 'logits_rowlse64']).max() < tol['logits']
        assert np.abs(logits.sum(1).numpy() - z[pre + 'logits_rowsum64']).max() < tol['logits'] * logits.shape[1] ** 0.5 * 4
        # queue state
        assert int(model.queue_ptr[0]) == int(z[pre + 'queue_ptr'])
        qn = model.queue[:, ptr0:ptr0 + N].cpu().numpy()
        assert np.abs(qn - z[pre + 'queue_new']).max() < tol['queue']
        assert abs(float(model.queue.double().sum()) - float(z[pre + 'queue_sum64'])) < tol['queue'] * N * 128
        # gradients, updated weights, EMA'd key weights, BN statistics
        qsd = dict(model.encoder_q.named_parameters())
        ksd = model.encoder_k.state_dict()
        for n in G.WATCH:
            g = qsd[n].grad.double().norm().item()
            ref = float(z[pre + 'gradnorm/' + n])
            assert abs(g - ref) <= tol['grad'] * max(ref, 1e-6), (s, n, g, ref)
            assert abs(qsd[n].double().norm().item() - float(z[pre + 'qnorm/' + n])) <= tol['param'] * float(z[pre + 'qnorm/' + n]) + 1e-7
            assert abs(ksd[n].double().norm().item() - float(z[pre + 'knorm/' + n])) <= tol['param'] * float(z[pre + 'knorm/' + n]) + 1e-7
        qst = model.encoder_q.state_dict()
        for n in G.WATCH_STATS:
            assert np.abs(qst[n][:8].cpu().numpy() - z[pre + 'qstat/' + n]).max() < tol['stat']
            assert np.abs(ksd[n][:8].cpu().numpy() - z[pre + 'kstat/' + n]).max() < tol['stat']


TOL_F32 = dict(loss=1e-3, logits=1e-3, queue=1e-3, grad=5e-3, param=1e-4, stat=1e-3, exact_acc=True)
# bf16 storage of activations/weights: ~3 significant digits per op through 53 layers
TOL_BF16 = dict(loss=6e-2, logits=1.5e-1, queue=3e-2, grad=2e-1, param=1e-3, stat=5e-2, exact_acc=False)


def test_golden_small_fp32():
    _run_against_golden('moco_v2_r50_small', torch.float32, 3, TOL_F32)


def test_golden_cfg1_fp32():
    """BASELINE configs[0]: N=32, 2x224^2, K=65536 — golden from the reference's own code."""
    _run_against_golden('moco_v2_r50_cfg1', torch.float32, 2, TOL_F32)


def test_golden_small_bf16():
    _run_against_golden('moco_v2_r50_small', torch.bfloat16, 3, TOL_BF16)


def test_golden_cfg1_bf16():
    _run_against_golden('moco_v2_r50_cfg1', torch.bfloat16, 2, TOL_BF16)


def test_live_oracle_fp32_three_steps():
    """Oracle on the host cores vs HIP path, N=8, 96x80 (non-square, odd tiles), K=512."""
    K, N = 512, 8
    oracle = MoCoOracle(K=K, seed=3, t_max=1000)
    model, opt, sched = U.build_product(K, torch.float32)
    sched.T_max = 1000
    U.load_oracle_state(model, oracle)
    model.train()
    gen = torch.Generator().manual_seed(99)
    for s in range(3):
        xq = torch.randn(N, 3, 96, 80, generator=gen)
        xk = torch.randn(N, 3, 96, 80, generator=gen)
        ref = oracle.train_step(xq, xk)
        out = U.product_step(model, opt, sched, xq.to(DEV), xk.to(DEV))
        assert abs(float(out['loss']) - float(ref['loss'])) < 1e-3
        assert abs(float(out['acc1']) - float(ref['acc1'])) < 1e-3
        assert (model.queue.cpu() - oracle.queue).abs().max() < 1e-3
        assert model._ptr == oracle.queue_ptr
        qsd = dict(model.encoder_q.named_parameters())
        worst = 0.0
        for n, g in ref['grads'].items():
            d = (qsd[n].grad.cpu() - g).norm().item() / max(g.norm().item(), 1e-8)
            worst = max(worst, d)
        assert worst < 2e-2, worst
        ksd = model.encoder_k.state_dict()
        for n in ('0.conv1.weight', '0.layer4.2.bn3._mean', '1.mlp.2.weight'):
            assert (ksd[n].cpu() - oracle.k[n]).abs().max() < 1e-4


def test_shuffle_bn_is_output_neutral():
    """SURVEY §3.1 note A: with frozen-statistics BN in the key encoder, batch shuffle changes no
    output (single process: randperm + gather)."""
    K, N = 256, 8
    oracle = MoCoOracle(K=K, seed=5, t_max=1000)
    outs = []
    for shuffle in (False, True):
        model, opt, sched = U.build_product(K, torch.float32)
        U.load_oracle_state(model, oracle)
        model.shuffle_bn = shuffle
        model.train()
        gen = torch.Generator().manual_seed(7)
        xq = torch.randn(N, 3, 64, 64, generator=gen).to(DEV)
        xk = torch.randn(N, 3, 64, 64, generator=gen).to(DEV)
        out = model(xq, xk)
        outs.append((float(out['loss']), model.queue.clone()))
    assert abs(outs[0][0] - outs[1][0]) < 1e-5
    assert (outs[0][1] - outs[1][1]).abs().max() < 1e-5


def test_key_encoder_has_no_grad_and_extract_mode():
    model, opt, sched = U.build_product(256, torch.bfloat16)
    assert all(not p.requires_grad for p in model.encoder_k.parameters())
    x = torch.randn(4, 3, 64, 64).to(DEV)
    feat = model(x, mode='extract')
    assert feat.shape == (4, 2048, 2, 2) and feat.dtype == torch.float32
    with pytest.raises(AssertionError):
        model(torch.randn(6, 3, 64, 64).to(DEV), torch.randn(6, 3, 64, 64).to(DEV))  # 256 % 6 != 0
