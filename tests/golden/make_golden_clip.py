"""Generate tests/golden/clip_*.npz by EXECUTING THE REFERENCE's CLIP sources
(passl_v110/modeling/architectures/CLIPWrapper.py, backbones/clip.py, backbones/vision_transformer.py,
heads/clip_head.py) on torch-CPU through the paddle shim (oracle/ref_runner.py); backward = torch
autograd over the reference's forward graph, AdamW = oracle.clip.CLIPOracle.apply_adamw (Paddle's
optimizer kernel is not in the reference tree).

    python tests/golden/make_golden_clip.py

Seed-defined inputs (regenerable without /root/reference):
    weights: oracle.clip.CLIPOracle(cfg, seed=0, text_std_cap=0.05);  per step: image ~ N(0,1), then
    text = oracle.clip.make_text from torch.Generator().manual_seed(4242)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import ref_runner                                  # noqa: E402
from oracle.clip import CLIPOracle, VIT_B_16, VIT_B_32, SMALL, make_text  # noqa: E402

CASES = {
    'clip_small': dict(cfg=SMALL, N=8, steps=3),
    # configs/clip/vit-b-32.yaml architecture (ViT-B/32: 50 image tokens; 77 text tokens, causal)
    'clip_vit_b32': dict(cfg=VIT_B_32, N=4, steps=2),
    # BASELINE.json configs[4] architecture: ViT-B/16 (197 image tokens) + the same text tower
    'clip_vit_b16': dict(cfg=VIT_B_16, N=4, steps=2),
}
SOLVER = dict(lr=1e-3, beta1=0.9, beta2=0.98, weight_decay=0.0005)
STD_CAP = 0.05
WATCH = ['visual.class_embedding', 'visual.positional_embedding', 'visual.proj',
         'visual.patch_embed.proj.weight', 'visual.norm_pre.weight', 'visual.blocks.0.attn.qkv.weight',
         'visual.blocks.1.mlp.fc2.bias', 'visual.norm_post.bias', 'transformer.blocks.0.attn.qkv.bias',
         'transformer.blocks.1.mlp.fc1.weight', 'transformer.blocks.1.attn.proj.weight',
         'token_embedding.weight', 'positional_embedding', 'ln_final.weight', 'text_projection',
         'logit_scale']


def run_case(name, cfg, N, steps):
    torch.manual_seed(0)
    oracle = CLIPOracle(cfg, seed=0, text_std_cap=STD_CAP, **SOLVER)
    model = ref_runner.build_reference_clip(cfg)
    ref_runner.load_clip_state(model, oracle)
    model.train()
    gen = torch.Generator().manual_seed(4242)
    R = cfg['image_resolution']
    out = {}
    for s in range(steps):
        image = torch.randn(N, 3, R, R, generator=gen)
        text = make_text(gen, N, cfg['context_length'], cfg['vocab_size'])
        for p in model.parameters():
            p.grad = None
        res = model(image, text)
        res['loss'].backward()
        ps = dict(model.model.named_parameters())
        grads = {n: ps[n].grad.detach().clone() for n in ps}
        oracle.st = {n: p.detach().clone() for n, p in model.model.state_dict().items()}
        oracle.apply_adamw(grads)
        with torch.no_grad():
            for n, p in model.model.state_dict().items():
                p.copy_(oracle.st[n])
        pre = 's%d_' % s
        for k in ('loss', 'img_loss', 'text_loss'):
            out[pre + k] = np.float64(res[k].item())
        for n in WATCH:
            out[pre + 'gradnorm/' + n] = np.float64(grads[n].double().norm().item())
            out[pre + 'pnorm/' + n] = np.float64(oracle.st[n].double().norm().item())
        print(name, 'step', s, 'loss %.6f' % out[pre + 'loss'])
    o64 = CLIPOracle(cfg, seed=0, dtype=torch.float64, text_std_cap=STD_CAP, **SOLVER)
    gen = torch.Generator().manual_seed(4242)
    for s in range(steps):
        image = torch.randn(N, 3, R, R, generator=gen)
        text = make_text(gen, N, cfg['context_length'], cfg['vocab_size'])
        r = o64.train_step(image.double(), text)
        pre = 's%d_f64_' % s
        for k in ('loss', 'img_loss', 'text_loss'):
            out[pre + k] = np.float64(float(r[k]))
        out[pre + 'image_logits'] = r['image_logits'].numpy().copy()
        out[pre + 'image_features'] = r['image_features'][:, :16].numpy().copy()
        out[pre + 'text_features'] = r['text_features'][:, :16].numpy().copy()
        for n in WATCH:
            out[pre + 'gradnorm/' + n] = np.float64(r['grads'][n].norm().item())
            out[pre + 'pnorm/' + n] = np.float64(o64.st[n].norm().item())
        print(name, 'f64 step', s, 'loss %.6f' % out[pre + 'loss'])
    out['meta'] = np.array([N, steps], dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)


if __name__ == '__main__':
    assert ref_runner.available(), 'needs /root/reference'
    for name in (sys.argv[1:] or list(CASES)):
        run_case(name, **CASES[name])
