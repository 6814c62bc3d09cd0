"""Shared helpers for the SimCLR whole-step tests: build the product SimCLR, load oracle state,
run the reference's hook sequence (OptimizerHook LARS branch + LRSchedulerHook) by hand."""
import torch

from passl_amd.hip import config as hip_config
from passl_amd.modeling import build_model
from passl_amd.solver.lr_scheduler import simclrCosineWarmup
from passl_amd.solver.optimizer import LarsMomentumOptimizer

MODEL_CFG = dict(
    name='SimCLR',
    backbone=dict(name='ResNetsimclr', depth=50),
    neck=dict(name='NonLinearNeckfc3', in_channels=2048, hid_channels=2048, out_channels=128,
              with_avg_pool=False),
    head=dict(name='SimCLRContrastiveHead', temperature=0.1),
)
SOLVER = dict(T=0.1, lr=4.0, warmup_steps=2, t_max=1000)      # tests/golden/make_golden_simclr.py


def build_product(dtype, device='gpu', solver=SOLVER):
    hip_config.set_device(device)
    hip_config.set_compute_dtype(dtype)
    torch.manual_seed(0)
    import copy
    cfg = copy.deepcopy(MODEL_CFG)
    cfg['head']['temperature'] = solver['T']
    model = build_model(cfg)
    sched = simclrCosineWarmup(solver['lr'], solver['warmup_steps'], solver['t_max'])
    opt = LarsMomentumOptimizer(sched, momentum=0.9, lars_weight_decay=1e-4,
                                parameter_list=list(model.parameters()),
                                exclude_from_weight_decay=['scale', 'offset', '.bias'])
    return model, opt, sched


@torch.no_grad()
def load_oracle_state(model, oracle):
    sd = {'encoder.%s' % n: t.detach() for n, t in oracle.st.items()}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all(m.startswith('backbone.') for m in missing), missing
    return model


def product_step(model, opt, sched, xq, xk):
    out = model(xq, xk, mode='train')
    opt.clear_gradients()
    out['loss'].backward()
    opt.minimize(out['loss'])
    sched.step()
    return out
