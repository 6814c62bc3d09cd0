"""A minimal torch-CPU backed stand-in for the ``paddle`` API surface that the
reference's MoCo-v2 path touches.  TEST INFRASTRUCTURE ONLY.

Purpose: ``import paddle`` is impossible in the build container (no wheel, no
network), so the reference's own Python (moco.py, contrastive_head.py,
base_neck.py, resnet.py over the vendored resnetimagenet.py, freeze.py,
registry.py) cannot run as shipped.  With this shim installed as
``sys.modules['paddle']`` those *unmodified* source files execute on torch-CPU,
which pins the oracle at the PASSL-Python level (control flow, parameter
iteration order, what is EMA'd, queue indexing, loss/accuracy formulae).

What the shim itself asserts about Paddle (= the [Paddle-semantics] list in
oracle/README.md): BatchNorm2D momentum 0.9 / eps 1e-5 / biased running var,
``_mean/_variance`` are non-trainable members of ``parameters()`` ordered
``weight, bias, _mean, _variance``; Linear weight is [in, out];
``F.normalize`` divides by max(norm, 1e-12); CrossEntropyLoss = mean of
-log_softmax at the label.

It monkey-patches a few ``torch.Tensor`` methods (``transpose(list)``,
``set_value``, ``stop_gradient``, ``cuda`` no-op) — run it in a process that
does nothing else (tests use a subprocess).
"""
import sys
import types

import numpy as np
import torch
import torch.nn.functional as TF

_T = torch.Tensor


# ------------------------------------------------------------------ Tensor
def _install_tensor_patches():
    if getattr(_T, '_paddle_shim', False):
        return
    _orig_transpose = _T.transpose

    def transpose(self, *args):
        if len(args) == 1 and isinstance(args[0], (list, tuple)):
            return self.permute(*args[0])
        return _orig_transpose(self, *args)

    def set_value(self, value):
        with torch.no_grad():
            self.copy_(torch.as_tensor(value, dtype=self.dtype))

    _orig_norm, _orig_argmax = _T.norm, _T.argmax

    def norm(self, p=2, axis=None, keepdim=False, dim=None):
        return _orig_norm(self, p=p, dim=axis if dim is None else dim, keepdim=keepdim)

    def argmax(self, axis=None, keepdim=False, dim=None):
        return _orig_argmax(self, dim=axis if dim is None else dim, keepdim=keepdim)

    def clip_(self, lo=None, hi=None):        # in dygraph an in-place clip of a parameter is allowed
        with torch.no_grad():
            return self.clamp_(lo, hi)

    _orig_copy = _T.copy_

    def copy_(self, src, *blocking, **kw):        # Tensor.copy_(src, blocking) in Paddle
        with torch.no_grad():
            return _orig_copy(self, src)
    _T.copy_ = copy_
    _T.norm, _T.argmax, _T.clip_ = norm, argmax, clip_
    _T._share_buffer_to = lambda self, other: None
    _T.transpose = transpose
    _T.set_value = set_value
    _T.stop_gradient = property(lambda s: not s.requires_grad,
                                lambda s, v: s.requires_grad_(not v) if s.is_leaf else None)
    # Parameter.trainable = False  <=>  stop_gradient = True  [Paddle-semantics]
    _T.trainable = property(lambda s: s.requires_grad,
                            lambda s, v: s.requires_grad_(bool(v)) if s.is_leaf else None)
    _T.cuda = lambda self, *a, **k: self
    _T.astype = lambda self, dt: self.to(_dtype(dt))
    # v2 optimizers (passl/optimizer/*.py): state is keyed by Parameter.name; dense gradients only
    import itertools
    _uid = itertools.count()

    def _get_name(self):
        if '_pd_name' not in self.__dict__:
            self.__dict__['_pd_name'] = 'generated_tensor_%d' % next(_uid)
        return self.__dict__['_pd_name']
    _T.name = property(_get_name, lambda self, v: self.__dict__.__setitem__('_pd_name', v))
    _T.is_selected_rows = lambda self: False

    def clear_gradient(self, set_to_zero=True):
        if self.grad is not None:
            if set_to_zero:
                self.grad.zero_()
            else:
                self.grad = None
    _T.clear_gradient = clear_gradient
    _T._paddle_shim = True


WIDEN_FLOAT32 = False      # golden generation in fp64: an explicit astype("float32") keeps the wide type


def _dtype(dt):
    if isinstance(dt, torch.dtype):
        return dt
    if dt == 'float32' and WIDEN_FLOAT32:
        return torch.float64
    return {'float32': torch.float32, 'float64': torch.float64, 'int64': torch.int64,
            'int32': torch.int32, 'bool': torch.bool, None: torch.float32}[dt]


def _set_default_dtype_hook():
    pass


# ------------------------------------------------------------------ nn
class Layer(torch.nn.Module):
    def sublayers(self, include_self=False):
        mods = list(self.modules())
        return mods if include_self else mods[1:]

    def create_parameter(self, shape, attr=None, dtype='float32', is_bias=False,
                         default_initializer=None):
        p = torch.nn.Parameter(torch.zeros(*shape, dtype=_dtype(dtype)))
        if default_initializer is not None:
            default_initializer(p)
        return p

    def add_parameter(self, name, parameter):
        self.register_parameter(name, parameter)
        return parameter

    def set_state_dict(self, sd):
        return self.load_state_dict(sd)


class Sequential(torch.nn.Sequential, Layer):
    pass


class ReLU(torch.nn.ReLU, Layer):
    pass


class Conv2D(Layer):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0,
                 dilation=1, groups=1, padding_mode='zeros', weight_attr=None,
                 bias_attr=None, data_format='NCHW'):
        super().__init__()
        assert groups == 1 and dilation == 1 and data_format == 'NCHW'
        k = kernel_size if isinstance(kernel_size, (tuple, list)) else (kernel_size,) * 2
        self._stride, self._padding = stride, padding
        self.weight = torch.nn.Parameter(torch.randn(out_channels, in_channels, *k) * 0.01)
        self.bias = None if bias_attr is False else torch.nn.Parameter(torch.zeros(out_channels))

    def forward(self, x):
        return TF.conv2d(x, self.weight, self.bias, self._stride, self._padding)


class _BatchNormBase(Layer):
    """paddle.nn.layer.norm._BatchNormBase: stats are non-trainable Parameters
    (so they appear in parameters()), order weight, bias, _mean, _variance."""

    def __init__(self, num_features, momentum=0.9, epsilon=1e-05, weight_attr=None,
                 bias_attr=None, data_format='NCHW', use_global_stats=None, name=None):
        super().__init__()
        self._momentum, self._epsilon = momentum, epsilon
        self._use_global_stats = use_global_stats
        # weight_attr / bias_attr = False: no affine parameter (BatchNorm1D(dim2, weight_attr=False, bias_attr=False),
        # passl/models/mocov3.py:149-151)
        self.weight = None if weight_attr is False else torch.nn.Parameter(torch.ones(num_features))
        self.bias = None if bias_attr is False else torch.nn.Parameter(torch.zeros(num_features))
        self._mean = torch.nn.Parameter(torch.zeros(num_features), requires_grad=False)
        self._variance = torch.nn.Parameter(torch.ones(num_features), requires_grad=False)

    def forward(self, x):
        dims = [d for d in range(x.dim()) if d != 1]
        shape = [1, -1] + [1] * (x.dim() - 2)
        use_global = self._use_global_stats if self._use_global_stats is not None \
            else (not self.training)
        if use_global:
            mean, var = self._mean, self._variance
        else:
            mean = x.mean(dim=dims)
            var = x.var(dim=dims, unbiased=False)
            with torch.no_grad():
                m = self._momentum
                self._mean.copy_(m * self._mean + (1 - m) * mean)
                self._variance.copy_(m * self._variance + (1 - m) * var)
        inv = torch.rsqrt(var + self._epsilon)
        if self.weight is not None:
            inv = inv * self.weight
        y = (x - mean.reshape(shape)) * inv.reshape(shape)
        return y if self.bias is None else y + self.bias.reshape(shape)


class BatchNorm2D(_BatchNormBase):
    pass


class BatchNorm1D(_BatchNormBase):
    pass


class BatchNorm(_BatchNormBase):
    pass


class SyncBatchNorm(_BatchNormBase):
    pass


class GroupNorm(Layer):
    pass


class MaxPool2D(Layer):
    def __init__(self, kernel_size, stride=None, padding=0):
        super().__init__()
        self.k, self.s, self.p = kernel_size, stride, padding

    def forward(self, x):
        return TF.max_pool2d(x, self.k, self.s, self.p)


class AdaptiveAvgPool2D(Layer):
    def __init__(self, output_size, data_format='NCHW'):
        super().__init__()
        self.o = output_size

    def forward(self, x):
        return TF.adaptive_avg_pool2d(x, self.o)


class Linear(Layer):
    def __init__(self, in_features, out_features, weight_attr=None, bias_attr=None, name=None):
        super().__init__()
        self.weight = torch.nn.Parameter(torch.randn(in_features, out_features) * 0.01)
        self.bias = None if bias_attr is False else torch.nn.Parameter(torch.zeros(out_features))

    def forward(self, x):
        y = x @ self.weight
        return y if self.bias is None else y + self.bias


class LayerNorm(Layer):
    def __init__(self, normalized_shape, epsilon=1e-05, weight_attr=None, bias_attr=None, name=None):
        super().__init__()
        n = normalized_shape if isinstance(normalized_shape, int) else normalized_shape[-1]
        self._epsilon = epsilon
        self.weight = torch.nn.Parameter(torch.ones(n))
        self.bias = torch.nn.Parameter(torch.zeros(n))

    def forward(self, x):
        return TF.layer_norm(x, (x.shape[-1],), self.weight, self.bias, self._epsilon)


class GELU(Layer):
    def __init__(self, approximate=False, name=None):
        super().__init__()
        self._approximate = approximate

    def forward(self, x):
        return TF.gelu(x, approximate='tanh' if self._approximate else 'none')


class Dropout(Layer):
    def __init__(self, p=0.5, axis=None, mode='upscale_in_train', name=None):
        super().__init__()
        self.p = p

    def forward(self, x):
        assert self.p == 0.0 or not self.training, 'shim: dropout > 0 is not on the oracle path'
        return x


class LayerList(torch.nn.ModuleList, Layer):
    pass


class Embedding(Layer):
    def __init__(self, num_embeddings, embedding_dim, padding_idx=None, sparse=False, weight_attr=None,
                 name=None):
        super().__init__()
        self.weight = torch.nn.Parameter(torch.randn(num_embeddings, embedding_dim))

    def forward(self, x):
        return self.weight[x]


class _Initializer(object):
    """Callable initialisers (paddle.nn.initializer.*): in dygraph `init(param)` fills in place.
    `reshape` returns a view sharing storage, so initialising a reshaped weight initialises it."""

    def __call__(self, t, block=None):
        with torch.no_grad():
            t.copy_(self.sample(t))
        return t


class _Constant(_Initializer):
    def __init__(self, value=0.0):
        self.value = value

    def sample(self, t):
        return torch.full_like(t, self.value)


class _TruncatedNormal(_Initializer):
    def __init__(self, mean=0.0, std=1.0, name=None):
        self.mean, self.std = mean, std

    def sample(self, t):
        return torch.fmod(torch.randn_like(t), 2.0) * self.std + self.mean


class _Normal(_Initializer):
    def __init__(self, mean=0.0, std=1.0, name=None):
        self.mean, self.std = mean, std

    def sample(self, t):
        return torch.randn_like(t) * self.std + self.mean


class _XavierUniform(_Initializer):
    def __init__(self, fan_in=None, fan_out=None, name=None):
        self.fan_in, self.fan_out = fan_in, fan_out

    def sample(self, t):
        fi = t.shape[0] if t.dim() == 2 else t.shape[1] * t[0][0].numel()
        fo = t.shape[1] if t.dim() == 2 else t.shape[0] * t[0][0].numel()
        a = (6.0 / (fi + fo)) ** 0.5
        return (torch.rand_like(t) * 2 - 1) * a


class _Noop(_Initializer):
    def __init__(self, *a, **k):
        pass

    def sample(self, t):
        return t


class CrossEntropyLoss(Layer):
    def forward(self, logits, labels):
        return TF.cross_entropy(logits, labels)


def _normalize(x, p=2, axis=1, epsilon=1e-12, name=None):
    n = x.pow(2).sum(dim=axis, keepdim=True).sqrt().clamp_min(epsilon)
    return x / n


# ------------------------------------------------------------------ module tree
def install():
    """Create and register the fake ``paddle`` package.  Idempotent."""
    if 'paddle' in sys.modules and getattr(sys.modules['paddle'], '_is_shim', False):
        return sys.modules['paddle']
    _install_tensor_patches()

    def mod(name):
        m = types.ModuleType(name)
        sys.modules[name] = m
        return m

    paddle = mod('paddle')
    paddle._is_shim = True
    paddle.__version__ = '0.0-torch-shim'
    paddle.Tensor = torch.Tensor
    paddle.no_grad = torch.no_grad
    paddle.randn = lambda shape, dtype=None: torch.randn(*shape)
    paddle.zeros = lambda shape, dtype='float32': torch.zeros(*shape, dtype=_dtype(dtype))
    paddle.ones = lambda shape, dtype='float32': torch.ones(*shape, dtype=_dtype(dtype))
    paddle.full = lambda shape, v, dtype=None: torch.full(tuple(shape), float(v))
    paddle.normal = lambda mean=0.0, std=1.0, shape=None: torch.randn(*shape) * std + mean
    paddle.uniform = lambda shape, dtype=None, min=-1.0, max=1.0: \
        torch.rand(*shape) * (max - min) + min
    paddle.numel = lambda t: torch.tensor(t.numel())
    paddle.to_tensor = lambda x, dtype=None, **k: torch.as_tensor(
        x, dtype=None if dtype is None else _dtype(dtype))
    paddle.concat = lambda xs, axis=0: torch.cat(list(xs), dim=axis)

    def matmul(x, y, transpose_x=False, transpose_y=False):
        if transpose_x:
            x = x.transpose(-1, -2)
        if transpose_y:
            y = y.transpose(-1, -2)
        return torch.matmul(x, y)
    paddle.matmul = matmul
    def arange(start, end=None, step=1, dtype=None):
        dt = _dtype(dtype) if dtype else None
        return torch.arange(start, dtype=dt) if end is None else torch.arange(start, end, step, dtype=dt)
    paddle.arange = arange
    paddle.einsum = torch.einsum
    paddle.rand = lambda shape, dtype=None: torch.rand(*shape)

    def create_parameter(shape, dtype='float32', name=None, attr=None, is_bias=False,
                         default_initializer=None):
        p = torch.nn.Parameter(torch.zeros(*shape, dtype=_dtype(dtype)))
        if default_initializer is not None:
            default_initializer(p)
        return p
    paddle.create_parameter = create_parameter
    paddle.reshape = lambda x, shape: x.reshape(list(shape))
    paddle.unsqueeze = lambda x, axis: x.unsqueeze(axis)
    paddle.argmax = lambda x, axis=None: x.argmax() if axis is None else x.argmax(dim=axis)

    class ParamAttr(object):
        def __init__(self, *a, **k):
            pass
    paddle.ParamAttr = ParamAttr
    paddle.sum = lambda x, axis=None, keepdim=False: x.sum() if axis is None \
        else x.sum(dim=axis, keepdim=keepdim)
    paddle.cast = lambda x, dt: x.to(_dtype(dt))
    paddle.flatten = lambda x, start_axis=0, stop_axis=-1: torch.flatten(x, start_axis, stop_axis)
    paddle.randperm = lambda n: torch.randperm(n)
    paddle.argsort = lambda x, axis=-1: torch.argsort(x, dim=axis)
    paddle.index_select = lambda x, index, axis=0: torch.index_select(x, axis, index)
    paddle.linspace = lambda start, stop, num, dtype=None: torch.linspace(float(start), float(stop), int(num))   # mae.py:234 (drop-path ladder)

    def assign(x, output=None):
        if output is None:
            return x.clone()
        with torch.no_grad():
            output.copy_(x)
        return output
    paddle.assign = assign

    nn = mod('paddle.nn')
    paddle.nn = nn
    for cls in (Layer, Sequential, ReLU, Conv2D, BatchNorm2D, BatchNorm1D, BatchNorm,
                SyncBatchNorm, GroupNorm, MaxPool2D, AdaptiveAvgPool2D, Linear,
                CrossEntropyLoss, LayerNorm, GELU, Dropout, LayerList, Embedding):
        setattr(nn, cls.__name__, cls)
    F = mod('paddle.nn.functional')
    nn.functional = F
    F.normalize = _normalize
    F.relu = TF.relu
    F.one_hot = lambda x, num_classes: TF.one_hot(x.long(), num_classes).float()
    F.softmax = lambda x, axis=-1: torch.softmax(x, dim=axis)
    F.log_softmax = lambda x, axis=-1: torch.log_softmax(x, dim=axis)
    # softmax_with_cross_entropy(soft_label=True): -sum(label*log_softmax) per row, shape [N,1]
    F.softmax_with_cross_entropy = lambda logits, label, soft_label=False, axis=-1: \
        -(label * torch.log_softmax(logits, dim=axis)).sum(dim=axis, keepdim=True)
    # kl_div: the kldiv_loss op has no gradient for `label`  [Paddle-semantics]
    F.kl_div = lambda input, label, reduction='mean': TF.kl_div(input, label.detach(),
                                                                reduction=reduction)
    init_mod = mod('paddle.nn.initializer')
    nn.initializer = init_mod
    for nm in ('XavierNormal', 'KaimingNormal', 'Uniform'):
        setattr(init_mod, nm, type(nm, (_Noop,), {}))
    init_mod.Normal = _Normal
    init_mod.Constant = _Constant
    init_mod.TruncatedNormal = _TruncatedNormal
    init_mod.XavierUniform = _XavierUniform
    layer = mod('paddle.nn.layer')
    nn.layer = layer
    norm = mod('paddle.nn.layer.norm')
    layer.norm = norm
    norm._BatchNormBase = _BatchNormBase
    # CLIP row: additive float masks pass through; bool masks become (m - 1) * 1e9
    tr = mod('paddle.nn.layer.transformer')
    layer.transformer = tr
    tr._convert_attention_mask = lambda m, dtype: ((m.to(dtype) - 1.0) * 1e9 if m.dtype == torch.bool
                                                   else m.to(dtype))
    tensor_mod = mod('paddle.tensor')
    paddle.tensor = tensor_mod
    tensor_mod.triu = lambda x, diagonal=0: torch.triu(x, diagonal)
    # v2 tree (passl/models/{vision_transformer,mocov3}.py, passl/nn/init.py, passl/models/utils/averaged_model.py)
    rnd = mod('paddle.tensor.random')
    tensor_mod.random = rnd
    rnd.gaussian = lambda shape, mean=0.0, std=1.0, dtype=None: torch.randn(*shape) * std + mean
    rnd.uniform = lambda shape, min=-1.0, max=1.0, dtype=None: torch.rand(*shape) * (max - min) + min
    paddle.float32, paddle.float64, paddle.int64, paddle.int32 = torch.float32, torch.float64, torch.int64, torch.int32
    paddle.float16, paddle.bfloat16 = torch.float16, torch.bfloat16
    paddle.sin, paddle.cos = torch.sin, torch.cos
    paddle.meshgrid = lambda *xs: torch.meshgrid(*xs, indexing='ij')       # paddle.meshgrid is 'ij'  [Paddle-semantics]
    paddle.is_floating_point = lambda t: t.is_floating_point()
    paddle.lerp = lambda a, b, w: torch.lerp(a, b, w)
    amp = mod('paddle.amp')
    paddle.amp = amp
    import contextlib as _ctx
    amp.auto_cast = lambda *a, **k: _ctx.nullcontext()
    nn.Identity = type('Identity', (torch.nn.Identity, Layer), {})
    nn.Tanh = type('Tanh', (torch.nn.Tanh, Layer), {})

    class CosineSimilarity(Layer):
        """paddle.nn.CosineSimilarity(axis, eps=1e-8) = F.cosine_similarity: sum(x1*x2) / max(|x1|*|x2|, eps)
        [Paddle-semantics: paddle/nn/functional/common.py cosine_similarity]."""

        def __init__(self, axis=1, eps=1e-8):
            super().__init__()
            self._axis, self._eps = axis, eps

        def forward(self, x1, x2):
            w12 = (x1 * x2).sum(self._axis)
            w1 = (x1 * x1).sum(self._axis)
            w2 = (x2 * x2).sum(self._axis)
            return w12 / (w1 * w2).sqrt().clamp_min(self._eps)
    nn.CosineSimilarity = CosineSimilarity
    Layer.named_sublayers = lambda self, prefix='', include_self=False: (
        (n, m) for n, m in self.named_modules(prefix=prefix) if include_self or m is not self)
    paddle.get_default_dtype = lambda: torch.get_default_dtype()
    paddle.shape = lambda x: list(x.shape)
    F.sigmoid = torch.sigmoid

    dist = mod('paddle.distributed')
    paddle.distributed = dist
    dist.get_world_size = lambda: 1
    dist.get_rank = lambda: 0

    def all_gather(out_list, t):
        out_list.append(t)
    dist.all_gather = all_gather

    class ParallelEnv:
        local_rank = 0
        nranks = 1
    dist.ParallelEnv = ParallelEnv

    fluid = mod('paddle.fluid')
    paddle.fluid = fluid
    fl = mod('paddle.fluid.layers')
    fluid.layers = fl
    # fluid.layers.l2_normalize: x / sqrt(sum(x^2) + eps)   [Paddle-semantics]
    fl.l2_normalize = lambda x, axis, epsilon=1e-12: x * torch.rsqrt(
        x.pow(2).sum(dim=axis, keepdim=True) + epsilon)

    def squeeze(x, axes):
        if not axes:
            return x.reshape([d for d in x.shape if d != 1] or [1]) if x.dim() > 2 else x
        for a in sorted(axes, reverse=True):
            x = x.squeeze(a)
        return x
    fl.squeeze = squeeze
    fl.split = lambda x, num_or_sections, dim=-1: list(torch.chunk(x, num_or_sections, dim=dim))
    fl.reduce_mean = lambda x, dim=None: x.mean() if dim is None else x.mean(dim=dim)

    def fl_accuracy(input, label, k=1):
        top = input.topk(k, dim=1).indices
        return (top == label.reshape(-1, 1)).any(dim=1).float().mean()
    fl.accuracy = fl_accuracy
    df = mod('paddle.fluid.data_feeder')
    fluid.data_feeder = df
    df.convert_dtype = lambda d: str(d).replace('torch.', '')

    utils = mod('paddle.utils')
    paddle.utils = utils
    dl = mod('paddle.utils.download')
    utils.download = dl
    dl.get_weights_path_from_url = lambda *a, **k: None

    # ---- v2 solver / loss / metric sources (passl/optimizer, passl/scheduler, passl/loss, passl/metric)
    paddle.norm = lambda x, p=2, axis=None: x.norm(p=p) if axis is None else x.norm(p=p, axis=axis)
    paddle.zeros_like = lambda x, dtype=None: torch.zeros_like(x, dtype=None if dtype is None else _dtype(dtype))
    paddle.add_n = lambda xs: sum(xs[1:], xs[0]) if isinstance(xs, (list, tuple)) else xs

    def cross_entropy(input, label, soft_label=False, axis=-1, reduction='mean'):
        """paddle.nn.functional.cross_entropy, hard labels ([N] or [N, 1]) or soft labels, mean over rows
        [Paddle-semantics]."""
        lsm = torch.log_softmax(input, dim=axis)
        if soft_label:
            loss = -(label * lsm).sum(dim=axis)
        else:
            loss = -lsm.gather(axis, label.reshape(-1, 1).long()).reshape(-1)
        return loss.mean() if reduction == 'mean' else (loss.sum() if reduction == 'sum' else loss)
    F.cross_entropy = cross_entropy
    F.label_smooth = lambda label, epsilon=0.1: (1.0 - epsilon) * label + epsilon / label.shape[-1]

    opt = mod('paddle.optimizer')
    paddle.optimizer = opt
    lr_mod = mod('paddle.optimizer.lr')
    opt.lr = lr_mod

    class LRScheduler(object):
        """paddle.optimizer.lr.LRScheduler (python/paddle/optimizer/lr.py, 2.4 line) [Paddle-semantics]: the
        constructor takes the first step(); step() advances last_epoch by one, step(epoch) sets it; last_lr caches
        get_lr() (or _get_closed_form_lr() for an explicit epoch); __call__ returns the cached value."""

        def __init__(self, learning_rate=0.1, last_epoch=-1, verbose=False):
            if not isinstance(learning_rate, (float, int)):
                raise TypeError('The type of learning rate must be float, but received {}'.format(
                    type(learning_rate)))
            self.base_lr = float(learning_rate)
            self.last_lr = float(learning_rate)
            self.last_epoch = last_epoch
            self.verbose = verbose
            self._var_name = None
            self.step()

        def __call__(self):
            return self.last_lr

        def step(self, epoch=None):
            if epoch is None:
                self.last_epoch += 1
                self.last_lr = self.get_lr()
            else:
                self.last_epoch = epoch
                if hasattr(self, '_get_closed_form_lr'):
                    self.last_lr = self._get_closed_form_lr()
                else:
                    self.last_lr = self.get_lr()

        def get_lr(self):
            raise NotImplementedError
    lr_mod.LRScheduler = LRScheduler

    class MultiStepDecay(LRScheduler):
        """paddle.optimizer.lr.MultiStepDecay: base_lr * gamma^(milestones passed) [Paddle-semantics]."""

        def __init__(self, learning_rate, milestones, gamma=0.1, last_epoch=-1, verbose=False):
            self.milestones, self.gamma = list(milestones), gamma
            super().__init__(learning_rate, last_epoch, verbose)

        def get_lr(self):
            return self.base_lr * self.gamma ** sum(1 for m in self.milestones if self.last_epoch >= m)
    lr_mod.MultiStepDecay = MultiStepDecay

    metric = mod('paddle.metric')
    paddle.metric = metric

    def metric_accuracy(input, label, k=1):
        """paddle.metric.accuracy: fraction of rows whose label is among the k largest scores (ties: lower index
        first, as top_k) [Paddle-semantics]."""
        top = input.topk(k, dim=1).indices
        return (top == label.reshape(-1, 1)).any(dim=1).float().mean()
    metric.accuracy = metric_accuracy

    vision = mod('paddle.vision')
    paddle.vision = vision
    models = mod('paddle.vision.models')
    vision.models = models
    resnet = mod('paddle.vision.models.resnet')
    models.resnet = resnet
    return paddle


def bind_vision_resnet(resnetimagenet_module):
    """paddle.vision.models.ResNet := the reference's vendored copy
    (passl_v110/modeling/backbones/resnetimagenet.py)."""
    models = sys.modules['paddle.vision.models']
    resnet = sys.modules['paddle.vision.models.resnet']
    for n in ('ResNet', 'BasicBlock', 'BottleneckBlock'):
        setattr(resnet, n, getattr(resnetimagenet_module, n))
    models.ResNet = resnetimagenet_module.ResNet
