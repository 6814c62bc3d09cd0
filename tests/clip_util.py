"""Shared helpers for the CLIP whole-step tests."""
import torch

from passl_amd.hip import config as hip_config
from passl_amd.modeling import build_model
from passl_amd.solver.optimizer import AdamW

SOLVER = dict(lr=1e-3, beta1=0.9, beta2=0.98, weight_decay=0.0005)      # tests/golden/make_golden_clip.py
STD_CAP = 0.05


def build_product(cfg, dtype, device='gpu'):
    hip_config.set_device(device)
    hip_config.set_compute_dtype(dtype)
    torch.manual_seed(0)
    arch = dict(name='CLIP', qkv_bias=True, pre_norm=True, proj=True, patch_bias=False)
    arch.update({k: cfg[k] for k in ('embed_dim', 'image_resolution', 'vision_layers', 'vision_width',
                                     'vision_patch_size', 'context_length', 'vocab_size', 'transformer_width',
                                     'transformer_heads', 'transformer_layers')})
    model = build_model(dict(name='CLIPWrapper', architecture=arch, head=dict(name='CLIPHead')))
    opt = AdamW(SOLVER['lr'], beta1=SOLVER['beta1'], beta2=SOLVER['beta2'], epsilon=1e-8,
                weight_decay=SOLVER['weight_decay'], parameters=list(model.parameters()))
    return model, opt


@torch.no_grad()
def load_oracle_state(model, oracle):
    sd = {'model.%s' % n: t.detach().float() for n, t in oracle.st.items()}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and not missing, (missing, unexpected)
    return model


def product_param(model, name):
    """Oracle / reference state_dict name -> the product parameter (raw matrices live in a Linear)."""
    sd = model.model.state_dict()
    return sd[name]


def product_grad(model, name):
    ps = dict(model.model.named_parameters())
    if name in ('visual.proj', 'text_projection'):
        name += '.weight'
    return ps[name].grad


def product_step(model, opt, image, text):
    out = model(image, text, mode='train')
    opt.clear_grad()
    out['loss'].backward()
    opt.step()
    return out
