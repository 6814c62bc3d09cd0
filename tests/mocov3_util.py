"""Shared helpers for the MoCo-v3 whole-step tests (tests/test_mocov3_gpu.py, tests/dp_worker.py)."""
from functools import partial

import torch

from passl_amd.hip import config as hip_config
from passl_amd.hip import nn as hnn
from passl_amd.solver.optimizer import AdamW

SOLVER = dict(lr=1e-4, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.1)      # tests/golden/make_golden_mocov3.py


def build_product(cfg, dtype, device='gpu', max_steps=10, T=0.2, base_momentum=0.99):
    from passl_amd.models.mocov3 import MoCoV3Pretrain, MoCoV3ViT
    from passl_amd.utils.infohub import runtime_info_hub
    hip_config.set_device(device)
    hip_config.set_compute_dtype(dtype)
    torch.manual_seed(0)
    runtime_info_hub.max_steps = max_steps
    enc = partial(MoCoV3ViT, stop_grad_conv1=True, img_size=cfg['img_size'], patch_size=cfg['patch_size'],
                  embed_dim=cfg['embed_dim'], depth=cfg['depth'], num_heads=cfg['num_heads'],
                  mlp_ratio=cfg['mlp_ratio'], qkv_bias=True, norm_layer=partial(hnn.LayerNorm, epsilon=1e-6))
    model = MoCoV3Pretrain(enc, dim=cfg['dim'], mlp_dim=cfg['mlp_dim'], T=T, base_momentum=base_momentum)
    opt = AdamW(SOLVER['lr'], betas=(SOLVER['beta1'], SOLVER['beta2']), eps=SOLVER['eps'],
                weight_decay=SOLVER['weight_decay'], parameters=list(model.parameters()))
    return model, opt


def mom_key(k):
    if k.startswith('base_encoder.'):
        return 'momentum_encoder.model.0.' + k[len('base_encoder.'):]
    return 'momentum_encoder.model.1.' + k[len('predictor.'):]


@torch.no_grad()
def load_oracle_state(model, oracle):
    sd = {k: t.detach().float() for k, t in oracle.st.items()}
    sd.update({mom_key(k): t.detach().float() for k, t in oracle.mom.items()})
    sd['momentum_encoder.steps'] = torch.tensor(oracle.steps, dtype=torch.int64)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and not missing, (missing, unexpected)
    assert model.momentum_encoder._steps == oracle.steps
    return model


def product_step(model, opt, x1, x2):
    loss = model([x1, x2])
    opt.clear_grad()
    loss.backward()
    opt.step()
    return loss
