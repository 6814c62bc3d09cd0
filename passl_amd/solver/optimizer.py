"""Momentum-SGD over the flat parameter arena (one HIP launch per step).

Registered under the reference's name ``Momentum`` (passl_v110/solver/optimizer.py:24 registers
paddle.optimizer.Momentum) with Paddle's constructor spelling
``Momentum(learning_rate, momentum=0.9, parameters=None, weight_decay=None, ...)``.  A float
``weight_decay`` is L2 decay folded into the gradient of EVERY parameter (configs/moco has no
exclusion list): g += wd*p; v = mu*v + g; p -= lr*v  — the rule restated in-tree at
passl/optimizer/momentum.py:150-158.
"""
import torch

from ..hip import ops, streams
from .builder import OPTIMIZERS
from .lr_scheduler import LRScheduler


def _load_flat_state(dst, sd, key):
    """Copy one flat optimizer-state vector from a checkpoint dict.  Values may be torch tensors or the
    numpy arrays the checkpoint pickle stores (hooks/checkpoint_hook.py); the arena layout is this
    code base's own (one vector per EncoderArena), so a size mismatch — e.g. a Paddle optimizer
    state with per-parameter accumulators — is reported instead of broadcasting garbage."""
    if key not in sd:
        raise KeyError('optimizer state has no %r (flat-arena layout expected; Paddle per-parameter '
                       'accumulator files are not interchangeable)' % key)
    src = torch.as_tensor(sd[key])
    if src.numel() != dst.numel():
        raise ValueError('optimizer state %r has %d elements, the arena holds %d'
                         % (key, src.numel(), dst.numel()))
    dst.copy_(src.reshape(dst.shape).to(dst.dtype))


def _grads_complete(arena):
    """Everything that writes this arena's gradients is ordered before the update kernel: outstanding gradient
    collectives (GradReducer.finish) and work on the side stream — weight gradients, the backward of a forked
    downsample branch (hip/streams.py)."""
    if arena.reducer is not None:
        arena.reducer.finish()
    if arena.grads is not None and arena.grads.is_cuda:
        streams.join(arena.grads.device)


class _DeviceHyper(object):
    """Step-dependent scalars of an optimizer — {lr, beta1^t, beta2^t, unused} — held in DEVICE memory and read by
    the update kernel when it runs (ops.*_dev): nothing about the schedule is frozen into a kernel launch, so the
    whole training step can be captured once in a HIP graph and replayed (hip/graph.py) while the reference's
    per-iteration schedule (passl_v110/hooks/lr_scheduler_hook.py:26-28) moves on.

    ``push_hyper()`` = one host-side update of the step state (``_host_values``) + one asynchronous H2D copy from a
    ring of pinned slots (the host may run several steps ahead of the GPU: a slot is re-used only after the copy
    that read it has completed).  ``step()`` calls it itself unless the caller already did (graph replay: the copy
    is stream-ordered in front of the graph launch, never part of the graph)."""

    _RING = 16

    def _init_hyper(self, device):
        self._hyper_dev = torch.zeros(4, dtype=torch.float32, device=device)
        self._hyper_host = None
        self._hyper_ev = [None] * self._RING
        self._hyper_slot = 0
        self._hyper_pushed = False

    def _host_values(self):
        """-> (lr, beta1^t, beta2^t); advances host-side step state (AdamW's t)."""
        return self.get_lr(), 0.0, 0.0

    def push_hyper(self):
        dev = self._hyper_dev
        vals = self._host_values()
        if dev.is_cuda:
            if self._hyper_host is None:
                self._hyper_host = torch.zeros(self._RING, 4, dtype=torch.float32).pin_memory()
            i = self._hyper_slot
            if self._hyper_ev[i] is not None:
                self._hyper_ev[i].synchronize()          # the copy that read this slot RING steps ago is done
            h = self._hyper_host[i]
            h[0], h[1], h[2] = vals
            dev.copy_(h, non_blocking=True)
            ev = self._hyper_ev[i] or torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dev.device))
            self._hyper_ev[i] = ev
            self._hyper_slot = (i + 1) % self._RING
        else:
            dev[0], dev[1], dev[2] = vals
        self._hyper_pushed = True

    def set_lr(self, value):
        """paddle.optimizer.Optimizer.set_lr: a fixed rate from now on (refused while a scheduler drives the rate, as
        Paddle does)."""
        if isinstance(self._learning_rate, LRScheduler):
            raise RuntimeError("optimizer's learning rate can't be LRScheduler when invoke this API, because this "
                               'will lead to conflict.')
        self._learning_rate = float(value)

    def _hyper_for_step(self):
        if not self._hyper_pushed:
            self.push_hyper()
        self._hyper_pushed = False
        return self._hyper_dev


@OPTIMIZERS.register()
class Momentum(_DeviceHyper):
    type = 'momentum'

    def __init__(self, learning_rate=0.001, momentum=0.9, parameters=None, use_nesterov=False,
                 weight_decay=None, grad_clip=None, multi_precision=False, rescale_grad=1.0,
                 name=None, use_master_param=None, lr_func=None):
        # v2 spelling (passl/optimizer/momentum.py:25-45): use_master_param asks for fp32 masters next to fp16
        # parameters — the arena's parameters ARE fp32 masters; lr_func (LRCallable groups) is not used by the recipes
        if use_nesterov:
            raise NotImplementedError('nesterov momentum is not used on the MoCo path')
        if grad_clip is not None or lr_func is not None:
            raise NotImplementedError('grad_clip / lr_func are not used on the MoCo path')
        self._learning_rate = learning_rate
        self._momentum = float(momentum)
        self._wd = float(weight_decay) if weight_decay else 0.0
        self._rescale = float(rescale_grad)
        self._parameter_list = [p for p in (parameters or []) if p.requires_grad]
        arenas = []
        for p in self._parameter_list:
            a = getattr(p, '_passl_arena', None)
            if a is None:
                raise NotImplementedError('Momentum optimises parameters that live in an '
                                          'EncoderArena (flat buffer); got a free tensor')
            if a not in arenas:
                arenas.append(a)
        for a in arenas:
            n_listed = sum(1 for p in self._parameter_list if p._passl_arena is a)
            if n_listed != len(a.param_slices):
                raise NotImplementedError('optimising a subset of an arena is not supported')
        self._arenas = arenas
        self._velocity = [torch.zeros_like(a.flat[:a.n_train]) for a in arenas]
        self.grad_scale = 1.0      # set by the DP reducer to 1/world_size (sum -> mean)
        self._init_hyper(arenas[0].device if arenas else torch.device('cpu'))

    # ---- paddle.optimizer API used by the hooks
    def get_lr(self):
        lr = self._learning_rate
        return float(lr()) if isinstance(lr, LRScheduler) else float(lr)

    def clear_grad(self, set_to_zero=True):
        for a in self._arenas:
            a.clear_grad()

    clear_gradients = clear_grad

    @torch.no_grad()
    def step(self):
        hyper = self._hyper_for_step()
        for a, v in zip(self._arenas, self._velocity):
            _grads_complete(a)
            ops.momentum_sgd_dev(a.flat[:a.n_train], a.grads, v, hyper, self._momentum, self._wd,
                                 self.grad_scale * self._rescale)

    def state_dict(self):
        sd = {'velocity_%d' % i: v.detach().cpu() for i, v in enumerate(self._velocity)}
        if isinstance(self._learning_rate, LRScheduler):
            sd['LR_Scheduler'] = self._learning_rate.state_dict()
        return sd

    def set_state_dict(self, sd):
        for i, v in enumerate(self._velocity):
            _load_flat_state(v, sd, 'velocity_%d' % i)
        if 'LR_Scheduler' in sd and isinstance(self._learning_rate, LRScheduler):
            self._learning_rate.set_state_dict(sd['LR_Scheduler'])


def _paddle_auto_names(arena):
    """Dygraph auto-generated parameter names ('conv2d_3.w_0', 'batch_norm2d_1.b_0', 'linear_0.w_0')
    in construction order — what LarsMomentumOptimizer matches ``exclude_from_weight_decay``
    against (param.name, not the state_dict key)  [Paddle-semantics]."""
    from ..hip import nn as hnn
    counters, names = {}, []
    for mod in arena.module.modules():
        own = [n for n, p in mod._parameters.items() if p is not None]
        if not own:
            continue
        if isinstance(mod, hnn.Conv2D):
            kind = 'conv2d'
        elif isinstance(mod, hnn.BatchNorm1D):
            kind = 'batch_norm1d'
        elif isinstance(mod, hnn._BatchNormBase):
            kind = 'batch_norm2d'
        elif isinstance(mod, hnn.Linear):
            kind = 'linear'
        else:
            kind = type(mod).__name__.lower()
        idx = counters.get(kind, 0)
        counters[kind] = idx + 1
        for n in own:
            names.append('%s_%d.%s' % (kind, idx, 'w_0' if n == 'weight' else 'b_0'))
    return names


@OPTIMIZERS.register()
class LarsMomentumOptimizer(_DeviceHyper):
    """paddle.fluid.optimizer.LarsMomentumOptimizer (registered by the reference at
    passl_v110/solver/optimizer.py:25, built with ``parameter_list=`` at solver/builder.py:198-201,
    driven through ``minimize(loss)`` / ``clear_gradients()`` by hooks/optimizer_hook.py:26-45)
    as a two-launch multi-tensor update over the flat arena (passl_hip_lars_momentum).

    Rule per parameter tensor (lars_momentum op)  [Paddle-semantics]:
        local_lr = lr * lars_coeff * |p| / (|g| + wd*|p| + epsilon)   if wd > 0, |p| > 0, |g| > 0
                 = lr                                                otherwise
        v = mu*v + local_lr*(g + wd*p);  p = p - v
    ``exclude_from_weight_decay``: wd = 0 for parameters whose *Paddle name* contains one of the
    strings — the yaml's ["scale","offset",".bias"] match none of the dygraph auto-names."""
    type = 'lars_momentum'

    def __init__(self, learning_rate, momentum, lars_coeff=0.001, lars_weight_decay=0.0005,
                 parameter_list=None, regularization=None, grad_clip=None, name=None,
                 exclude_from_weight_decay=None, epsilon=0, multi_precision=False,
                 rescale_grad=1.0):
        if regularization is not None or grad_clip is not None:
            raise NotImplementedError('regularization / grad_clip are not used by configs/simclr')
        self._learning_rate = learning_rate
        self._momentum = float(momentum)
        self._coeff = float(lars_coeff)
        self._wd = float(lars_weight_decay)
        self._eps = float(epsilon)
        self._rescale = float(rescale_grad)
        self._exclude = list(exclude_from_weight_decay or [])
        params = [p for p in (parameter_list or []) if p.requires_grad]
        arenas = []
        for p in params:
            a = getattr(p, '_passl_arena', None)
            if a is None:
                raise NotImplementedError('LarsMomentumOptimizer optimises parameters that live in '
                                          'an EncoderArena (flat buffer); got a free tensor')
            if a not in arenas:
                arenas.append(a)
        for a in arenas:
            if sum(1 for p in params if p._passl_arena is a) != len(a.param_slices):
                raise NotImplementedError('optimising a subset of an arena is not supported')
        self._parameter_list = params
        self._arenas = arenas
        self._velocity = [torch.zeros_like(a.flat[:a.n_train]) for a in arenas]
        self._tables = [self._build_table(a) for a in arenas]
        self.grad_scale = 1.0
        self._init_hyper(arenas[0].device if arenas else torch.device('cpu'))

    def _build_table(self, arena, chunk=4096):
        names = _paddle_auto_names(arena)
        assert len(names) == len(arena.param_slices)
        blk_off, blk_len, blk_seg, seg_wd = [], [], [], []
        self.param_names = names
        for si, ((off, n), name) in enumerate(zip(arena.param_slices, names)):
            seg_wd.append(0.0 if any(e in name for e in self._exclude) else self._wd)
            for c in range(0, n, chunk):
                blk_off.append(off + c)
                blk_len.append(min(chunk, n - c))
                blk_seg.append(si)
        dev = arena.device
        return dict(blk_off=torch.tensor(blk_off, dtype=torch.int64, device=dev),
                    blk_len=torch.tensor(blk_len, dtype=torch.int32, device=dev),
                    blk_seg=torch.tensor(blk_seg, dtype=torch.int32, device=dev),
                    seg_wd=torch.tensor(seg_wd, dtype=torch.float32, device=dev),
                    norms=torch.zeros(len(seg_wd) + len(blk_len), 2, dtype=torch.float32, device=dev))

    def get_lr(self):
        lr = self._learning_rate
        return float(lr()) if isinstance(lr, LRScheduler) else float(lr)

    def clear_gradients(self, set_to_zero=True):
        for a in self._arenas:
            a.clear_grad()

    clear_grad = clear_gradients

    @torch.no_grad()
    def step(self):
        hyper = self._hyper_for_step()
        for a, v, t in zip(self._arenas, self._velocity, self._tables):
            _grads_complete(a)
            ops.lars_momentum_dev(a.flat[:a.n_train], a.grads, v, t, hyper, self._momentum, self._coeff,
                                  self._eps, self.grad_scale * self._rescale)

    def minimize(self, loss=None, startup_program=None, parameters=None, no_grad_set=None):
        """Dygraph ``minimize``: the gradients already exist (the hook called backward())."""
        self.step()

    def state_dict(self):
        sd = {'velocity_%d' % i: v.detach().cpu() for i, v in enumerate(self._velocity)}
        if isinstance(self._learning_rate, LRScheduler):
            sd['LR_Scheduler'] = self._learning_rate.state_dict()
        return sd

    def set_state_dict(self, sd):
        for i, v in enumerate(self._velocity):
            _load_flat_state(v, sd, 'velocity_%d' % i)
        if 'LR_Scheduler' in sd and isinstance(self._learning_rate, LRScheduler):
            self._learning_rate.set_state_dict(sd['LR_Scheduler'])


@OPTIMIZERS.register()
class MomentumLARC(LarsMomentumOptimizer):
    """passl/optimizer/momentum_larc.py:25-111 (the optimizer of the SimSiam linear-probe recipe,
    tasks/ssl/simsiam/configs/simsiam_resnet50_lp_in1k_1n8c_dp_fp32.yaml) as the multi-tensor LARS machinery with the
    LARC update (passl_hip_larc_momentum_dev).  Rule per parameter tensor:
        if |p| != 0 and |g| != 0:  a = trust_coefficient |p| / (|g| + |p| wd + eps)   [clip: a = min(a / lr, 1)]
                                   g = a (g + wd p)
        v = mu v + g;  p -= lr v          (a zero-norm tensor takes its raw gradient, without weight decay)
    Unlike LARS the learning rate multiplies the velocity, not the gradient: a schedule rescales the whole history.
    ``use_master_param``: parameters are fp32 masters here anyway."""
    type = 'momentum_larc'

    def __init__(self, learning_rate=0.0, momentum=0.9, weight_decay=0.0, trust_coefficient=0.02, clip=True,
                 eps=1e-8, use_master_param=True, grad_clip=None, parameters=None, lr_func=None, **args):
        if grad_clip is not None or lr_func is not None:
            raise NotImplementedError('grad_clip / lr_func are not used by the linear-probe recipes')
        super().__init__(learning_rate, momentum, lars_coeff=trust_coefficient, lars_weight_decay=weight_decay,
                         parameter_list=parameters, epsilon=eps)
        self._clip = bool(clip)

    @torch.no_grad()
    def step(self):
        hyper = self._hyper_for_step()
        for a, v, t in zip(self._arenas, self._velocity, self._tables):
            _grads_complete(a)
            ops.larc_momentum_dev(a.flat[:a.n_train], a.grads, v, t, hyper, self._momentum, self._coeff,
                                  self._eps, self._clip, self.grad_scale * self._rescale)


@OPTIMIZERS.register()
class AdamW(_DeviceHyper):
    """paddle.optimizer.AdamW (registered by the reference at passl_v110/solver/optimizer.py:22) as ONE
    launch over the flat arena.  adamw op  [Paddle-semantics]:
        p *= 1 - lr*wd;  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;
        p -= lr*sqrt(1-b2^t)/(1-b1^t) * m / (sqrt(v) + eps*sqrt(1-b2^t))
    A float ``weight_decay`` decays EVERY trainable parameter (configs/mae/mae_vit_b_pretrain.yaml has no
    exclusion list; fixed sin-cos embeddings are buffers and are not touched)."""
    type = 'adamw'

    def __init__(self, learning_rate=0.001, beta1=0.9, beta2=0.999, epsilon=1e-08, parameters=None,
                 weight_decay=0.01, lr_ratio=None, apply_decay_param_fun=None, grad_clip=None,
                 lazy_mode=False, multi_precision=False, name=None, betas=None, eps=None,
                 use_master_param=None, exp_avg_force_fp32=None):
        # v2 spelling (passl/optimizer/adamw.py:24-50, the MoCo-v3 yaml): betas=(b1, b2), eps.  use_master_param /
        # exp_avg_force_fp32 ask for fp32 master weights and an fp32 first moment next to fp16 parameters: the
        # flat arena IS fp32 (weights and both moments; the compute-dtype copy is derived from it every step), so
        # both are always satisfied and the flags are accepted as no-ops.
        if betas is not None:
            beta1, beta2 = (float(b) for b in betas)
        if eps is not None:
            epsilon = eps
        if apply_decay_param_fun is not None or lr_ratio is not None or grad_clip is not None:
            raise NotImplementedError('per-parameter decay / lr ratios / clipping are not used by the MAE '
                                      'pre-training config and are not built')
        self._learning_rate = learning_rate
        self._b1, self._b2, self._eps = float(beta1), float(beta2), float(epsilon)
        self._wd = float(weight_decay) if weight_decay else 0.0
        params = [p for p in (parameters or []) if p.requires_grad]
        arenas = []
        for p in params:
            a = getattr(p, '_passl_arena', None)
            if a is None:
                raise NotImplementedError('AdamW optimises parameters that live in an EncoderArena')
            if a not in arenas:
                arenas.append(a)
        for a in arenas:
            if sum(1 for p in params if p._passl_arena is a) != len(a.param_slices):
                raise NotImplementedError('optimising a subset of an arena is not supported')
        self._parameter_list = params
        self._arenas = arenas
        self._m = [torch.zeros_like(a.flat[:a.n_train]) for a in arenas]
        self._v = [torch.zeros_like(a.flat[:a.n_train]) for a in arenas]
        self._t = 0
        self.grad_scale = 1.0
        self._init_hyper(arenas[0].device if arenas else torch.device('cpu'))

    def _host_values(self):
        self._t += 1
        return self.get_lr(), self._b1 ** self._t, self._b2 ** self._t

    def get_lr(self):
        lr = self._learning_rate
        return float(lr()) if isinstance(lr, LRScheduler) else float(lr)

    def clear_grad(self, set_to_zero=True):
        for a in self._arenas:
            a.clear_grad()

    clear_gradients = clear_grad

    @torch.no_grad()
    def step(self):
        hyper = self._hyper_for_step()
        for a, m, v in zip(self._arenas, self._m, self._v):
            _grads_complete(a)
            ops.adamw_dev(a.flat[:a.n_train], a.grads, m, v, hyper, self._b1, self._b2, self._eps, self._wd,
                          self.grad_scale)

    def state_dict(self):
        sd = {'t': self._t}
        for i, (m, v) in enumerate(zip(self._m, self._v)):
            sd['moment1_%d' % i] = m.detach().cpu()
            sd['moment2_%d' % i] = v.detach().cpu()
        if isinstance(self._learning_rate, LRScheduler):
            sd['LR_Scheduler'] = self._learning_rate.state_dict()
        return sd

    def set_state_dict(self, sd):
        self._t = int(sd['t'])
        for i, (m, v) in enumerate(zip(self._m, self._v)):
            _load_flat_state(m, sd, 'moment1_%d' % i)
            _load_flat_state(v, sd, 'moment2_%d' % i)
        if 'LR_Scheduler' in sd and isinstance(self._learning_rate, LRScheduler):
            self._learning_rate.set_state_dict(sd['LR_Scheduler'])
