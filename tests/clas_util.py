"""Shared helpers for the linear-probe whole-step tests."""
import torch

from passl_amd.hip import config as hip_config
from passl_amd.modeling import build_model
from passl_amd.solver.optimizer import Momentum

LR, MU = 0.002, 0.9                                   # tests/golden/make_golden_clas.py


def build_product(num_classes, dtype, device='gpu', frozen_stages=4):
    hip_config.set_device(device)
    hip_config.set_compute_dtype(dtype)
    torch.manual_seed(0)
    model = build_model(dict(name='Classification', backbone=dict(name='ResNet', depth=50,
                                                                    frozen_stages=frozen_stages),
                             head=dict(name='ClasHead', with_avg_pool=True, in_channels=2048,
                                       num_classes=num_classes)))
    opt = Momentum(LR, momentum=MU, parameters=list(model.parameters()), weight_decay=0.0)
    return model, opt


@torch.no_grad()
def load_oracle_state(model, oracle):
    sd = {n: t.detach().float() for n, t in oracle.st.items()}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and not missing, (missing, unexpected)
    return model


def product_step(model, opt, img, labels):
    out = model(img, labels, mode='train')
    opt.clear_grad()
    out['loss'].backward()
    opt.step()
    return out
