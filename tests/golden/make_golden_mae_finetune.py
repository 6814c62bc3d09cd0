"""Generate tests/golden/mae_ft_*.npz by EXECUTING THE REFERENCE's fine-tuning sources — MAE_FINETUNE
(passl_v110/modeling/architectures/MAE.py:58-94) over MAE_ViT (backbones/mae.py:190-314: class token + learnable
position table, pre-norm blocks, global average pool of the patch tokens, fc_norm) and VisionTransformerClsHead
(heads/vision_transformer_head.py:23-60) — built by the reference's registries from the `model:` block of
configs/mae/mae_vit_b_finetune.yaml, on torch-CPU through the paddle shim (oracle/ref_runner.py).  Backward = torch
autograd over the reference's forward graph; AdamW = oracle.mae.MAEOracle.apply_adamw with the yaml's betas / decay.

    python tests/golden/make_golden_mae_finetune.py

Seed-defined inputs (regenerable without /root/reference): weights = finetune_state() below over the model's own
state_dict keys (recorded in the fixture); per step imgs ~ N(0,1), labels ~ U{0..classes-1} from
torch.Generator().manual_seed(909)."""
import copy
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import ref_runner                      # noqa: E402
from oracle.mae import MAEOracle, finetune_state   # noqa: E402

SOLVER = dict(lr=1e-3, beta1=0.9, beta2=0.999, weight_decay=0.05)        # mae_vit_b_finetune.yaml:58-62
ARCH_YAML = dict(name='MAE_ViT', patch_size=16, embed_dim=768, depth=12, num_heads=12, qkv_bias=True, mlp_ratio=4)
CASES = {
    'mae_ft_small': dict(arch=dict(ARCH_YAML, embed_dim=128, depth=2, num_heads=4, img_size=64), classes=16, N=8, steps=3),
    # the yaml's architecture as written (ViT-B/16, 224^2: 197 tokens), 1000 classes, tiny batch
    'mae_ft_vit_b': dict(arch=dict(ARCH_YAML), classes=1000, N=2, steps=2),
}
WATCH = ['backbone.cls_token', 'backbone.pos_embed', 'backbone.patch_embed.proj.weight', 'backbone.blocks.0.attn.qkv.weight',
         'backbone.blocks.1.mlp.fc2.bias', 'backbone.blocks.1.norm2.weight', 'backbone.fc_norm.weight',
         'backbone.fc_norm.bias', 'head.fc_cls.weight', 'head.fc_cls.bias']


def run_case(name, arch, classes, N, steps):
    torch.manual_seed(0)
    ns = ref_runner.load()
    cfg = dict(name='MAE_FINETUNE', architecture=copy.deepcopy(arch),
               head=dict(name='VisionTransformerClsHead', num_classes=classes, in_channels=arch['embed_dim']))
    model = ns.build_model(cfg)
    sd = model.state_dict()
    keys_shapes = [(k, tuple(v.shape)) for k, v in sd.items()]
    st = finetune_state(keys_shapes)
    with torch.no_grad():
        for k, v in sd.items():
            v.copy_(st[k])
    model.train()
    opt = MAEOracle(dict(img_size=32, patch_size=16, embed_dim=32, depth=1, decoder_embed_dim=32, decoder_depth=1,
                         mlp_ratio=1.0), **SOLVER)              # (its AdamW rule only; the state comes from the model)
    hw = arch.get('img_size', 224)
    gen = torch.Generator().manual_seed(909)
    feats = {}
    model.head.register_forward_pre_hook(lambda mod, args: feats.update(x=args[0].detach().clone()))
    out = {}
    for s in range(steps):
        x = torch.randn(N, 3, hw, hw, generator=gen)
        y = torch.randint(0, classes, (N,), generator=gen)
        for p in model.parameters():
            p.grad = None
        scores = {}
        h = model.head.fc_cls.register_forward_hook(lambda mod, a, o: scores.update(s=o.detach().clone()))
        res = model(x, y, mode='train')
        h.remove()
        res['loss'].backward()
        ps = dict(model.named_parameters())
        grads = {n: ps[n].grad.detach().clone() for n in ps if ps[n].grad is not None}
        opt.st = {n: p.detach().clone() for n, p in model.state_dict().items()}
        opt.apply_adamw(grads)
        with torch.no_grad():
            for n, p in model.state_dict().items():
                p.copy_(opt.st[n])
        pre = 's%d_' % s
        out[pre + 'loss'] = np.float64(res['loss'].item())
        out[pre + 'acc1'] = np.float64(float(res['acc1']))
        out[pre + 'acc5'] = np.float64(float(res['acc5']))
        out[pre + 'feat_head'] = feats['x'][:, :8].numpy().copy()
        out[pre + 'score_head'] = scores['s'][:, :8].numpy().copy()
        for n in WATCH:
            out[pre + 'gradnorm/' + n] = np.float64(grads[n].double().norm().item())
            out[pre + 'pnorm/' + n] = np.float64(opt.st[n].double().norm().item())
        print(name, 'step', s, 'loss %.6f acc1 %.1f acc5 %.1f' % (out[pre + 'loss'], out[pre + 'acc1'], out[pre + 'acc5']))
    out['meta'] = np.array([N, hw, steps, classes], dtype=np.int64)
    out['keys'] = np.array(['%s:%s' % (k, 'x'.join(map(str, s_))) for k, s_ in keys_shapes])
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)


if __name__ == '__main__':
    assert ref_runner.available(), 'needs /root/reference'
    for name in (sys.argv[1:] or list(CASES)):
        run_case(name, **CASES[name])
