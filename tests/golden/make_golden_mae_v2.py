"""Generate tests/golden/mae_v2_*.npz by EXECUTING THE REFERENCE's v2 MAE sources — passl/models/mae.py
(MaskedAutoencoderViT :37-290, factory mae_vit_base_patch16_dec512d8b :331-345) over
passl/models/vision_transformer.py (Attention :116-156, Block :159-206, PatchEmbed :209-249) and
passl/models/utils/pos_embed.py — on torch-CPU through the paddle shim, in a process of their own
(oracle/ref_runner_v2.py: the v2 tree imports itself as ``passl``).  Backward = torch autograd over the reference's
forward graph; AdamW = oracle.mae.MAEOracle.apply_adamw (the optimizer kernel is not in the reference tree).

    python tests/golden/make_golden_mae_v2.py

Seed-defined inputs (regenerable without /root/reference): weights oracle.mae.MAEOracle(cfg, seed=0); per step
imgs ~ N(0,1), then noise ~ U[0,1) from torch.Generator().manual_seed(4242) (the noise replaces paddle.rand in
random_masking).  Also stored: what the reference's OWN initialisation produces (initialize_weights :117-151), as
per-tensor statistics — the product's constructor is held to them.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import ref_runner_v2                   # noqa: E402
from oracle.mae import MAEOracle, VIT_B            # noqa: E402

SMALL = dict(img_size=64, patch_size=16, embed_dim=128, depth=2, num_heads=4, decoder_embed_dim=64,
             decoder_depth=2, decoder_num_heads=2, mlp_ratio=4.0)
CASES = {
    'mae_v2_small': dict(cfg=SMALL, N=4, steps=3, norm_pix_loss=True, factory=None),
    # the factory the v2 recipes name (tasks/ssl/mae): ViT-B/16 encoder, 512-wide 8-block decoder
    'mae_v2_vit_b': dict(cfg=VIT_B, N=2, steps=2, norm_pix_loss=True, factory='mae_vit_base_patch16_dec512d8b'),
}
SOLVER = dict(lr=1e-3, beta1=0.9, beta2=0.95, weight_decay=0.05)
WATCH = ['patch_embed.proj.weight', 'cls_token', 'mask_token', 'blocks.0.attn.qkv.weight',
         'blocks.1.mlp.fc2.bias', 'blocks.1.norm2.weight', 'norm.bias', 'decoder_embed.weight',
         'decoder_blocks.0.attn.proj.weight', 'decoder_blocks.1.mlp.fc1.weight', 'decoder_pred.bias']
INIT_STATS = ['patch_embed.proj.weight', 'patch_embed.proj.bias', 'cls_token', 'mask_token', 'pos_embed',
              'decoder_pos_embed', 'blocks.0.attn.qkv.weight', 'blocks.0.attn.qkv.bias', 'blocks.1.mlp.fc1.weight',
              'blocks.0.norm1.weight', 'blocks.0.norm1.bias', 'decoder_embed.weight', 'decoder_pred.weight',
              'decoder_pred.bias']


def run_case(ns, name, cfg, N, steps, norm_pix_loss, factory):
    torch.manual_seed(0)
    full = dict(cfg, norm_pix_loss=norm_pix_loss)
    oracle = MAEOracle(full, seed=0, **SOLVER)
    if factory is not None:
        model = getattr(ns.mae, factory)(norm_pix_loss=norm_pix_loss)
    else:
        from functools import partial
        import paddle.nn as pnn
        model = ns.mae.MaskedAutoencoderViT(norm_layer=partial(pnn.LayerNorm, epsilon=1e-6),
                                            norm_pix_loss=norm_pix_loss, **cfg)
    out = {}
    sd = model.state_dict()
    for n in INIT_STATS:                 # the reference's own initialisation, before the seed-defined state goes in
        t = sd[n].detach().double()
        out['init/' + n] = np.array([t.mean().item(), t.std(unbiased=False).item(), t.abs().max().item()])
    out['init_pos_embed_head'] = sd['pos_embed'][0, :3, :8].detach().numpy().copy()
    assert set(sd.keys()) == set(oracle.st.keys()), set(sd.keys()) ^ set(oracle.st.keys())
    with torch.no_grad():
        for n, t in oracle.st.items():
            assert sd[n].shape == t.shape, (n, sd[n].shape, t.shape)
            sd[n].copy_(t.detach())
    model.train()
    paddle = sys.modules['paddle']
    gen = torch.Generator().manual_seed(4242)
    L = (cfg['img_size'] // cfg['patch_size']) ** 2
    for s in range(steps):
        x = torch.randn(N, 3, cfg['img_size'], cfg['img_size'], generator=gen)
        noise = torch.rand(N, L, generator=gen)
        paddle.rand = lambda shape, dtype=None: noise.clone()
        for p in model.parameters():
            p.grad = None
        loss, pred, mask = model(x, 0.75)
        loss.backward()
        ps = dict(model.named_parameters())
        grads = {n: ps[n].grad.detach().clone() for n in ps if ps[n].grad is not None}
        oracle.st = {n: p.detach().clone() for n, p in model.state_dict().items()}
        oracle.apply_adamw({n: grads[n] for n in oracle.st if n in grads})
        with torch.no_grad():
            for n, p in model.state_dict().items():
                p.copy_(oracle.st[n])
        pre = 's%d_' % s
        out[pre + 'loss'] = np.float64(loss.item())
        out[pre + 'mask'] = mask.numpy().astype(np.uint8)
        out[pre + 'pred_head'] = pred[:, :4, :8].detach().numpy().copy()
        out[pre + 'pred_sum64'] = np.float64(pred.double().sum().item())
        for n in WATCH:
            out[pre + 'gradnorm/' + n] = np.float64(grads[n].double().norm().item())
            out[pre + 'pnorm/' + n] = np.float64(oracle.st[n].double().norm().item())
        print(name, 'step', s, 'loss %.6f' % out[pre + 'loss'])
    o64 = MAEOracle(full, seed=0, dtype=torch.float64, **SOLVER)
    gen = torch.Generator().manual_seed(4242)
    for s in range(steps):
        x = torch.randn(N, 3, cfg['img_size'], cfg['img_size'], generator=gen)
        noise = torch.rand(N, L, generator=gen)
        r = o64.train_step(x.double(), noise.double())
        pre = 's%d_f64_' % s
        out[pre + 'loss'] = np.float64(float(r['loss']))
        out[pre + 'pred_head'] = r['pred'][:, :4, :8].numpy().copy()
        for n in WATCH:
            out[pre + 'gradnorm/' + n] = np.float64(r['grads'][n].norm().item())
            out[pre + 'pnorm/' + n] = np.float64(o64.st[n].norm().item())
        print(name, 'f64 step', s, 'loss %.6f' % out[pre + 'loss'])
    out['meta'] = np.array([N, steps, int(norm_pix_loss)], dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)


if __name__ == '__main__':
    assert ref_runner_v2.available(), 'needs /root/reference'
    ns = ref_runner_v2.load_mae()
    for name in (sys.argv[1:] or list(CASES)):
        run_case(ns, name, **CASES[name])
