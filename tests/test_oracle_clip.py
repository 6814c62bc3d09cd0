"""CPU tests that pin the CLIP oracle (oracle/clip.py): against golden vectors produced by running
the reference's own CLIP sources (tests/golden/make_golden_clip.py), live against those sources when
/root/reference is present, and known answers for the causal mask / EOT selection / loss."""
import math
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import clip as C

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')
SOLVER = dict(lr=1e-3, beta1=0.9, beta2=0.98, weight_decay=0.0005)
WATCH = ['visual.class_embedding', 'visual.positional_embedding', 'visual.proj',
         'visual.patch_embed.proj.weight', 'visual.norm_pre.weight', 'visual.blocks.0.attn.qkv.weight',
         'visual.blocks.1.mlp.fc2.bias', 'visual.norm_post.bias', 'transformer.blocks.0.attn.qkv.bias',
         'transformer.blocks.1.mlp.fc1.weight', 'transformer.blocks.1.attn.proj.weight',
         'token_embedding.weight', 'positional_embedding', 'ln_final.weight', 'text_projection',
         'logit_scale']
# see tests/test_clip_gpu.py: zero-gradient bias slice + Adam = noise-defined parameter norm
ADAM_NOISE_DEFINED = {'transformer.blocks.0.attn.qkv.bias'}


def _against(name, cfg, max_steps):
    z = np.load(os.path.join(GOLDEN, name + '.npz'))
    N, steps = [int(v) for v in z['meta']]
    o = C.CLIPOracle(cfg, seed=0, text_std_cap=0.05, **SOLVER)
    gen = torch.Generator().manual_seed(4242)
    R = cfg['image_resolution']
    for s in range(min(steps, max_steps)):
        image = torch.randn(N, 3, R, R, generator=gen)
        text = C.make_text(gen, N, cfg['context_length'], cfg['vocab_size'])
        out = o.train_step(image, text)
        pre = 's%d_' % s
        for k in ('loss', 'img_loss', 'text_loss'):
            assert abs(float(out[k]) - float(z[pre + k])) < 5e-5, (k, float(out[k]), float(z[pre + k]))
        np.testing.assert_allclose(out['image_logits'].numpy(), z['s%d_f64_image_logits' % s], atol=5e-4)
        for n in WATCH:
            g = out['grads'][n].double().norm().item()
            assert abs(g - float(z[pre + 'gradnorm/' + n])) <= 5e-4 * max(g, 1e-9), n
            if n not in ADAM_NOISE_DEFINED:
                assert abs(o.st[n].double().norm().item() - float(z[pre + 'pnorm/' + n])) < 1e-4, n


def test_oracle_matches_golden_small():
    _against('clip_small', C.SMALL, 3)


def test_oracle_matches_golden_vit_b32_first_step():
    _against('clip_vit_b32', C.VIT_B_32, 1)


@pytest.mark.skipif(not os.path.isdir('/root/reference/passl_v110'),
                    reason='reference tree not present (GPU box)')
def test_oracle_matches_reference_sources_live():
    """fp64 on both sides (paddle_shim.WIDEN_FLOAT32 keeps ln_final's explicit float32 cast wide); the
    text blocks keep the reference's own initial std (width^-0.5 * 2 depth)."""
    code = r'''
import torch
torch.set_default_dtype(torch.float64)
from oracle import ref_runner, paddle_shim, clip as oc
paddle_shim.WIDEN_FLOAT32 = True
cfg = dict(oc.SMALL, context_length=9, vocab_size=120, transformer_layers=1, vision_layers=1)
m = ref_runner.build_reference_clip(cfg)
o = oc.CLIPOracle(cfg, seed=3, dtype=torch.float64)
ref_runner.load_clip_state(m, o)
g = torch.Generator().manual_seed(5)
img = torch.randn(5, 3, 64, 64, generator=g, dtype=torch.float64); text = oc.make_text(g, 5, 9, 120)
out = m(img, text)
out['loss'].backward()
r = o.train_step(img, text)
for k in ('loss', 'img_loss', 'text_loss'):
    assert abs(float(out[k].detach()) - float(r[k])) < 1e-12, k
ps = dict(m.model.named_parameters())
assert set(ps) == set(r['grads'])
for n, gr in r['grads'].items():
    assert (ps[n].grad - gr).abs().max().item() <= 1e-11 * max(gr.abs().max().item(), 1.0), n
# in-place clip of logit_scale after the logits (clip.py:309-311)
o.st['logit_scale'] = torch.full((1,), 5.0, dtype=torch.float64)
ref_runner.load_clip_state(m, o)                      # the oracle took an AdamW step above
out2 = m(img, text); r2 = o.train_step(img, text)
assert abs(float(m.model.logit_scale) - 4.6) < 1e-12 and abs(float(out2['loss'].detach()) - float(r2['loss'])) < 1e-10
print('LIVE-OK')
'''
    r = subprocess.run([sys.executable, '-c', code], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'LIVE-OK' in r.stdout, r.stderr[-3000:]


def test_known_answers_mask_eot_loss():
    # causal mask: token t never sees later tokens -> changing the last token leaves earlier rows alone
    cfg = dict(C.SMALL, context_length=6, vocab_size=50, transformer_layers=1)
    st = C.init_state(torch.Generator().manual_seed(0), cfg, text_std_cap=0.05)
    text = torch.tensor([[5, 7, 49, 0, 0, 0], [3, 49, 0, 0, 0, 0]])
    t2 = text.clone()
    t2[:, 5] = 11
    assert torch.allclose(C.encode_text(st, text, cfg), C.encode_text(st, t2, cfg), atol=0)   # EOT rows at 2 / 1
    t3 = text.clone()
    t3[0, 0] = 9                                          # an EARLIER token does change the EOT feature
    assert (C.encode_text(st, text, cfg)[0] - C.encode_text(st, t3, cfg)[0]).abs().max() > 1e-6
    # EOT = first maximum
    assert torch.tensor([[1, 9, 9, 0]]).argmax(dim=-1).item() == 1
    # loss: identical unit features, scale s -> logits s*I; CE = log(1 + (B-1) e^-s) on both sides
    B, s = 4, math.log(1 / 0.07)
    eye = torch.eye(B, 8)
    sc = math.exp(s)
    il = sc * eye @ eye.t()
    ce = torch.nn.functional.cross_entropy(il, torch.arange(B))
    assert abs(float(ce) - math.log(1 + (B - 1) * math.exp(-sc))) < 1e-6
    # make_text: EOT is the row maximum, padding zeros after it
    tx = C.make_text(torch.Generator().manual_seed(1), 16, 12, 300)
    am = tx.argmax(dim=-1)
    assert bool((tx[torch.arange(16), am] == 299).all())
    assert all(int(tx[b, am[b] + 1:].abs().sum()) == 0 for b in range(16))
