"""Generate tests/golden/mae_*.npz by EXECUTING THE REFERENCE's MAE sources
(passl_v110/modeling/backbones/mae.py class MAE, modules/get_sincos_pe.py) on torch-CPU through the
paddle shim (oracle/ref_runner.py); backward = torch autograd over the reference's forward graph,
AdamW = oracle.mae.MAEOracle.apply_adamw (Paddle's optimizer kernel is not in the reference tree).

    python tests/golden/make_golden_mae.py

Seed-defined inputs (regenerable without /root/reference):
    weights: oracle.mae.MAEOracle(cfg, seed=0);  per step: imgs ~ N(0,1), then noise ~ U[0,1)
    from torch.Generator().manual_seed(777)  (the noise replaces paddle.rand in random_masking)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import ref_runner                      # noqa: E402
from oracle.mae import MAEOracle, VIT_B            # noqa: E402

SMALL = dict(img_size=64, patch_size=16, embed_dim=128, depth=2, num_heads=4, decoder_embed_dim=64,
             decoder_depth=2, decoder_num_heads=2, mlp_ratio=4.0)
CASES = {
    'mae_small': dict(cfg=SMALL, N=4, steps=3, norm_pix_loss=True),
    'mae_small_rawpix': dict(cfg=SMALL, N=6, steps=1, norm_pix_loss=False),
    # BASELINE.json configs[3] architecture (ViT-B/16, mask 0.75: 50 / 197 tokens), tiny batch
    'mae_vit_b': dict(cfg=VIT_B, N=2, steps=2, norm_pix_loss=True),
}
SOLVER = dict(lr=1e-3, beta1=0.9, beta2=0.95, weight_decay=0.05)
WATCH = ['patch_embed.proj.weight', 'cls_token', 'mask_token', 'blocks.0.attn.qkv.weight',
         'blocks.1.mlp.fc2.bias', 'blocks.1.norm2.weight', 'norm.bias', 'decoder_embed.weight',
         'decoder_blocks.0.attn.proj.weight', 'decoder_blocks.1.mlp.fc1.weight', 'decoder_pred.bias']


def run_case(name, cfg, N, steps, norm_pix_loss):
    torch.manual_seed(0)
    full = dict(cfg, norm_pix_loss=norm_pix_loss)
    oracle = MAEOracle(full, seed=0, **SOLVER)
    model = ref_runner.build_reference_mae(cfg, norm_pix_loss=norm_pix_loss)
    ref_runner.load_mae_state(model, oracle)
    model.train()
    paddle = sys.modules['paddle']
    gen = torch.Generator().manual_seed(777)
    L = (cfg['img_size'] // cfg['patch_size']) ** 2
    out = {}
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
    gen = torch.Generator().manual_seed(777)
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
    assert ref_runner.available(), 'needs /root/reference'
    for name in (sys.argv[1:] or list(CASES)):
        run_case(name, **CASES[name])
