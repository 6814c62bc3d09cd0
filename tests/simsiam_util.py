"""Shared helpers for the SimSiam whole-step tests."""
import torch

from passl_amd.hip import config as hip_config
from passl_amd.solver.optimizer import Momentum

SOLVER = dict(lr=2e-4, predictor_lr=5e-4, momentum=0.9, weight_decay=1e-4)       # tests/golden/make_golden_simsiam.py


class TwoGroups(object):
    """The task yaml's parameter groups: encoder on the schedule, predictor at its fixed rate."""

    def __init__(self, model, lr, predictor_lr, momentum, weight_decay):
        enc = [p for n, p in model.named_parameters() if p.requires_grad and n.startswith('encoder')]
        pred = [p for n, p in model.named_parameters() if p.requires_grad and n.startswith('predictor')]
        self.enc = Momentum(lr, momentum=momentum, parameters=enc, weight_decay=weight_decay)
        self.pred = Momentum(predictor_lr, momentum=momentum, parameters=pred, weight_decay=weight_decay)

    def clear_grad(self):
        self.enc.clear_grad()
        self.pred.clear_grad()

    def step(self):
        self.enc.step()
        self.pred.step()


def build_product(dtype, device='gpu'):
    from passl_amd.models import build_model
    hip_config.set_device(device)
    hip_config.set_compute_dtype(dtype)
    torch.manual_seed(0)
    model = build_model(dict(name='simsiam_resnet50_pretrain'))
    return model, TwoGroups(model, **SOLVER)


@torch.no_grad()
def load_oracle_state(model, oracle):
    sd = {k: t.detach().float() for k, t in oracle.st.items()}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and not missing, (missing, unexpected)
    return model


def product_step(model, opt, x1, x2):
    loss = model([x1, x2])
    opt.clear_grad()
    loss.backward()
    opt.step()
    return loss
