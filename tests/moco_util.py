"""Shared helpers for the whole-step tests: build the product MoCo, load oracle state into it,
run the reference's hook sequence (OptimizerHook + LRSchedulerHook) by hand."""
import copy

import torch

from passl_amd.hip import config as hip_config
from passl_amd.modeling import build_model
from passl_amd.solver.lr_scheduler import CosineAnnealingDecay
from passl_amd.solver.optimizer import Momentum

MODEL_CFG = dict(
    name='MoCo',
    backbone=dict(name='ResNet', depth=50),
    neck=dict(name='NonLinearNeckV1', in_channels=2048, hid_channels=2048, out_channels=128,
              with_avg_pool=True),
    head=dict(name='ContrastiveHead', temperature=0.2),
)


def build_product(K, dtype, device='gpu', v1=False):
    """v1: the `model:` / solver blocks of configs/moco/moco_v1_r50.yaml (LinearNeck, T = 0.07, lr 0.03
    MultiStepDecay [120, 160] epochs x 5004 iterations)."""
    hip_config.set_device(device)
    hip_config.set_compute_dtype(dtype)
    cfg = copy.deepcopy(MODEL_CFG)
    cfg.update(K=K)
    if v1:
        cfg['neck'] = dict(name='LinearNeck', in_channels=2048, out_channels=128, with_avg_pool=True)
        cfg['head']['temperature'] = 0.07
    torch.manual_seed(0)
    model = build_model(cfg)
    if v1:
        from passl_amd.solver.lr_scheduler import MultiStepDecay
        sched = MultiStepDecay(0.03, milestones=[120 * 5004, 160 * 5004])
        opt = Momentum(sched, parameters=list(model.parameters()), weight_decay=1e-4)
        return model, opt, sched
    sched = CosineAnnealingDecay(0.015, T_max=200 * 5004)
    opt = Momentum(sched, parameters=list(model.parameters()), weight_decay=1e-4)
    return model, opt, sched


@torch.no_grad()
def load_oracle_state(model, oracle):
    sd = {}
    for enc, st in (('encoder_q', oracle.q), ('encoder_k', oracle.k)):
        for n, t in st.items():
            sd['%s.%s' % (enc, n)] = t.detach()
    sd['queue'] = oracle.queue
    sd['queue_ptr'] = torch.tensor([oracle.queue_ptr], dtype=torch.int64)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all(m.startswith('backbone.') for m in missing), missing   # alias of encoder_q.0
    return model


def product_step(model, opt, sched, xq, xk):
    """OptimizerHook.train_iter_end + LRSchedulerHook.train_iter_end."""
    out = model(xq, xk, mode='train', total_iters=1, current_iter=1, mixup_fn=None)
    opt.clear_grad()
    out['loss'].backward()
    opt.step()
    sched.step()
    return out
