"""CPU tests that pin the MAE oracle (oracle/mae.py): against golden vectors produced by running
the reference's own MAE sources (tests/golden/make_golden_mae.py), live against those sources when
/root/reference is present, and known answers for masking / sin-cos embedding / AdamW / schedule."""
import math
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import mae as M

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')
SMALL = dict(img_size=64, patch_size=16, embed_dim=128, depth=2, num_heads=4, decoder_embed_dim=64,
             decoder_depth=2, decoder_num_heads=2, mlp_ratio=4.0)
SOLVER = dict(lr=1e-3, beta1=0.9, beta2=0.95, weight_decay=0.05)
WATCH = ['patch_embed.proj.weight', 'cls_token', 'mask_token', 'blocks.0.attn.qkv.weight',
         'blocks.1.mlp.fc2.bias', 'blocks.1.norm2.weight', 'norm.bias', 'decoder_embed.weight',
         'decoder_blocks.0.attn.proj.weight', 'decoder_blocks.1.mlp.fc1.weight', 'decoder_pred.bias']


def _against(name, cfg, max_steps):
    z = np.load(os.path.join(GOLDEN, name + '.npz'))
    N, steps, npl = [int(v) for v in z['meta']]
    o = M.MAEOracle(dict(cfg, norm_pix_loss=bool(npl)), seed=0, **SOLVER)
    gen = torch.Generator().manual_seed(777)
    L = (cfg['img_size'] // cfg['patch_size']) ** 2
    for s in range(min(steps, max_steps)):
        x = torch.randn(N, 3, cfg['img_size'], cfg['img_size'], generator=gen)
        noise = torch.rand(N, L, generator=gen)
        out = o.train_step(x, noise)
        pre = 's%d_' % s
        assert abs(float(out['loss']) - float(z[pre + 'loss'])) < 2e-6
        assert np.array_equal(out['mask'].numpy().astype(np.uint8), z[pre + 'mask'])
        np.testing.assert_allclose(out['pred'][:, :4, :8].numpy(), z[pre + 'pred_head'], atol=5e-6)
        for n in WATCH:
            g = out['grads'][n].double().norm().item()
            assert abs(g - float(z[pre + 'gradnorm/' + n])) <= 1e-4 * max(g, 1e-9), n
            assert abs(o.st[n].double().norm().item() - float(z[pre + 'pnorm/' + n])) < 1e-5, n


def test_oracle_matches_golden_small():
    _against('mae_small', SMALL, 3)
    _against('mae_small_rawpix', SMALL, 1)


def test_oracle_matches_golden_vit_b_first_step():
    _against('mae_vit_b', M.VIT_B, 1)


@pytest.mark.skipif(not os.path.isdir('/root/reference/passl_v110'),
                    reason='reference tree not present (GPU box)')
def test_oracle_matches_reference_sources_live():
    code = r'''
import sys, torch
from oracle import ref_runner
from oracle.mae import MAEOracle
cfg = dict(img_size=48, patch_size=8, embed_dim=64, depth=1, num_heads=2, decoder_embed_dim=64,
           decoder_depth=1, decoder_num_heads=2, mlp_ratio=4.0, norm_pix_loss=True)
o = MAEOracle(cfg, seed=4)
m = ref_runner.build_reference_mae(cfg, norm_pix_loss=True)
ref_runner.load_mae_state(m, o)
m.train()
g = torch.Generator().manual_seed(3)
x = torch.randn(3, 3, 48, 48, generator=g); noise = torch.rand(3, 36, generator=g)
sys.modules['paddle'].rand = lambda shape, dtype=None: noise.clone()
loss, pred, mask = m(x, 0.75)
loss.backward()
r = o.train_step(x, noise)
assert abs(float(loss.detach()) - float(r['loss'])) < 1e-6
assert (pred - r['pred']).abs().max().item() < 5e-6 and bool((mask == r['mask']).all())
ps = dict(m.named_parameters())
assert ps['pos_embed'].grad is None and ps['decoder_pos_embed'].grad is None
for n, gr in r['grads'].items():
    assert (ps[n].grad - gr).abs().max().item() <= 5e-6 * max(gr.abs().max().item(), 1.0), n
print('LIVE-OK')
'''
    r = subprocess.run([sys.executable, '-c', code], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'LIVE-OK' in r.stdout, r.stderr[-3000:]


def test_masking_posembed_adamw_known_answers():
    noise = torch.tensor([[0.9, 0.1, 0.5, 0.3, 0.7, 0.2, 0.8, 0.4]])
    keep, mask, restore = M.random_masking_ids(noise, 0.75)
    assert keep.tolist() == [[1, 5]] and mask.tolist() == [[1, 0, 1, 1, 1, 0, 1, 1]]
    assert restore.tolist() == [[7, 0, 4, 2, 5, 1, 6, 3]]          # rank of each patch's noise
    pe = M.sincos_2d(8, 2)
    assert pe.shape == (5, 8) and np.allclose(pe[0], 0) and np.allclose(pe[1], [0, 0, 1, 1, 0, 0, 1, 1])
    # w goes first: patch (h=0, w=1) encodes w in the FIRST half
    assert np.allclose(pe[2][:4], [math.sin(1.0), math.sin(0.01), math.cos(1.0), math.cos(0.01)])
    # AdamW single step: decoupled decay then bias-corrected Adam
    o = M.MAEOracle(SMALL, seed=0, lr=0.1, weight_decay=0.5)
    p0 = o.st['norm.weight'].clone()
    g = torch.full_like(p0, 2.0)
    o.apply_adamw({'norm.weight': g})
    m, v = 0.1 * 2.0, 0.05 * 4.0
    lr_t = 0.1 * math.sqrt(1 - 0.95) / (1 - 0.9)
    exp = p0 * (1 - 0.1 * 0.5) - lr_t * m / (math.sqrt(v) + 1e-8 * math.sqrt(1 - 0.95))
    assert (o.st['norm.weight'] - exp).abs().max() < 1e-6
    # schedule of configs/mae/mae_vit_b_pretrain.yaml in iterations (x iters_per_epoch = 10)
    f = lambda t: M.warmup_cosine_lr(t, 3.75e-5, 8000, 1e-5, 400, 1e-6, 1e-3)
    assert f(0) == 1e-6 and abs(f(200) - (1e-6 + (1e-3 - 1e-6) / 2)) < 1e-12 and abs(f(400) - 3.75e-5) < 1e-12
    assert abs(f(4400) - (1e-5 + (3.75e-5 - 1e-5) / 2)) < 1e-12
