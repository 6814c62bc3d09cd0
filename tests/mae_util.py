"""Shared helpers for the MAE whole-step tests."""
import torch

from passl_amd.hip import config as hip_config
from passl_amd.modeling import build_model
from passl_amd.solver.optimizer import AdamW

SOLVER = dict(lr=1e-3, beta1=0.9, beta2=0.95, weight_decay=0.05)      # tests/golden/make_golden_mae.py


def build_product(cfg, dtype, norm_pix_loss, device='gpu'):
    hip_config.set_device(device)
    hip_config.set_compute_dtype(dtype)
    torch.manual_seed(0)
    arch = dict(name='MAE', img_size=cfg['img_size'], patch_size=cfg['patch_size'], embed_dim=cfg['embed_dim'],
                depth=cfg['depth'], num_heads=cfg['num_heads'], decoder_embed_dim=cfg['decoder_embed_dim'],
                decoder_depth=cfg['decoder_depth'], decoder_num_heads=cfg['decoder_num_heads'],
                mlp_ratio=cfg['mlp_ratio'], norm_pix_loss=norm_pix_loss)
    model = build_model(dict(name='MAE_PRETRAIN', architecture=arch))
    opt = AdamW(SOLVER['lr'], beta1=SOLVER['beta1'], beta2=SOLVER['beta2'],
                weight_decay=SOLVER['weight_decay'], parameters=list(model.parameters()))
    return model, opt


@torch.no_grad()
def load_oracle_state(model, oracle):
    sd = {'backbone.%s' % n: t.detach().float() for n, t in oracle.st.items()}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and not missing, (missing, unexpected)
    return model


def product_step(model, opt, imgs, noise):
    out = model(imgs, mode='train', noise=noise)
    opt.clear_grad()
    out['loss'].backward()
    opt.step()
    return out
