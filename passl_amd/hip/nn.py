"""Layers of the MoCo-v2 R50 hot path, executed by libpassl_hip.so.

PyTorch provides parameter storage and the autograd tape only: every ``autograd.Function``
below launches hand-written HIP kernels through the C ABI (passl_amd/hip/ops.py) in both
directions.  Activations are NHWC in the compute dtype (bf16, or fp32 for parity runs).

Storage model (EncoderArena): all parameters and BN statistics of one encoder live in ONE flat
fp32 buffer, each conv weight physically [K][R][S][C] and each Linear weight [out][in]
(K-contiguous rows = the B operand of the implicit GEMM); the ``nn.Parameter`` objects are
strided *views* with the reference's logical shapes ([Cout,Cin,kh,kw], Linear [in,out]) so
``state_dict()`` matches the reference layout.  Gradients are written by the
kernels straight into a flat fp32 gradient buffer (``p.grad`` are views of it), so the key-
encoder EMA, the optimizer and the DP all-reduce are single launches over flat memory.

Reference semantics mirrored here: paddle.nn.Conv2D(bias_attr=False), BatchNorm2D (momentum 0.9,
eps 1e-5, biased running var, ``_use_global_stats``), MaxPool2D(3,2,1), AdaptiveAvgPool2D(1),
Linear — as used by passl_v110/modeling/backbones/resnetimagenet.py:111-253 and
passl_v110/modeling/necks/base_neck.py:68-97.
"""
from types import SimpleNamespace

import torch
import torch.nn as tnn
from torch.autograd import Function

from . import config, ops, streams
from . import plan as P
from .packer import WeightPacker


class Layer(tnn.Module):
    """torch Module with the few paddle.nn.Layer spellings the reference's code relies on."""

    def sublayers(self, include_self=False):
        mods = list(self.modules())
        return mods if include_self else mods[1:]

    def set_state_dict(self, sd, *a, **k):
        return self.load_state_dict(sd, *a, **k)


def _need_rt(layer):
    if getattr(layer, '_rt', None) is None:
        raise RuntimeError('%s is not attached to an EncoderArena (call EncoderArena(encoder) after '
                           'building the model)' % type(layer).__name__)
    return layer._rt


class GradSlot:
    """Hand-off of ONE gradient tensor between two backward nodes of a residual fork.

    A block input x feeds conv1 and the identity (or downsample) branch, so autograd would add the
    two input gradients with an extra elementwise kernel.  Instead the branch whose backward runs
    first (bn3's identity gradient, or the downsample conv's dx) `put`s its tensor here and
    returns None to autograd; conv1's backward `take`s it and the data-gradient kernel adds it in
    its epilogue.  The order is fixed by the graph (conv1's backward depends on bn3's), and it is
    checked: a `take` from a slot that was armed but never filled raises instead of silently
    dropping a gradient."""

    __slots__ = ('armed', 'grad', 'ready')

    def __init__(self):
        self.armed = False
        self.grad = None
        self.ready = None

    def arm(self):
        self.armed = True

    def put(self, g):
        if not self.armed or self.grad is not None:
            raise RuntimeError('GradSlot.put: slot not armed or already filled')
        self.grad = g
        # the two ends may run on different streams (a downsample branch on the side stream,
        # hip/streams.py): the taker waits for the producer's launches
        self.ready = streams.record_event(torch.cuda.current_stream(g.device)) if g.is_cuda else None

    def take(self):
        if not self.armed:
            return None
        if self.grad is None:
            raise RuntimeError('GradSlot.take: the residual-branch gradient has not been produced '
                               'yet (autograd executed the fork in an unexpected order)')
        g, self.grad, self.armed = self.grad, None, False
        if self.ready is not None:
            # g may come from the side stream's pool (a forked downsample branch): ordered by the event; its
            # block is recycled behind later side-stream work only, all of which waits on main-stream events
            # recorded after this launch (hip/streams.py, "Memory") — no record_stream
            streams.wait_event(torch.cuda.current_stream(g.device), self.ready)
            self.ready = None
        return g


class BNLink:
    """Hand-off between a training-mode BatchNorm(+ReLU) layer and the ONE convolution that consumes
    its output.  The BatchNorm's forward records what its backward statistics need (its input y, the
    batch statistics, the ReLU mask source); the consumer's data-gradient launch — which produces
    exactly the gradient w.r.t. the BatchNorm output — masks that gradient and writes the per-tile
    (sum g, sum g*xhat) slab in its epilogue (csrc/igemm_epi.h), and reports it here.  The BatchNorm's
    backward then skips its reduce pass, and the gradient it receives is already g."""

    __slots__ = ('y', 'st', 'mask', 'relu_mode', 'fused', 'res_link')

    def __init__(self, y, st, mask, relu_mode, res_link=None):
        self.y, self.st, self.mask, self.relu_mode = y, st, mask, relu_mode
        self.fused = None            # (slab, tiles) once a data-gradient launch has done the work
        # the BNLink of the BatchNorm (without ReLU) whose output is this layer's RESIDUAL input and has no other
        # consumer — a bottleneck block's downsample branch: relu(bn3(y3) + bn_ds(y_ds)).  The masked gradient g of
        # this layer's output is then the output gradient of that BatchNorm as well, and the data-gradient launch that
        # reduces (sum g, sum g*xhat) for this layer can do it for that one in the same pass (conv desc bnb2_*).
        self.res_link = res_link


def bn_link(t):
    """The BNLink riding on a BatchNorm output tensor (None for anything else)."""
    return getattr(t, '_passl_bn_link', None)


class WgradLink:
    """Hand-off in the other direction, for the stem: the layer that PRODUCES a convolution's output gradient (the fused
    BatchNorm + ReLU + max-pool backward, two launches from the end of a backward pass) launches that convolution's
    weight gradient itself, piece of the batch by piece, each on the side stream as soon as the apply pass has written
    the piece (`launch`), and says so (`done`); the convolution's own backward then has nothing left to launch."""

    __slots__ = ('launch', 'done')

    def __init__(self):
        self.launch = None           # launch(dy, n0, n1, i, parts): images n0..n1 of dy (piece i) are written; set by the convolution
        self.done = False            # set by the producer of dy once every piece has been handed over


def wgrad_link(t):
    """The WgradLink riding on a convolution output tensor (None for anything else)."""
    return getattr(t, '_passl_wgrad_link', None)


# =============================================================================== conv
def _stem_wgrad(layer, pl, x, dy, tmp):
    """The stem filter is padded to [K][7][8 taps][4 channels] for the kernel; `tmp` (zeroed) collects its gradient."""
    ops.conv_wgrad(pl.wd, x, dy.reshape(-1, layer.geom.cout), tmp)


def _stem_wgrad_finish(layer, tmp):
    """... which goes back into the dense [K][7][7][3] slot of the flat gradient buffer."""
    ops.unpad_add(tmp, layer._rt.dw, layer.geom.cout, 7, 7, 3, P.STEM_ROW // 4, 4)


class _ConvFn(Function):
    @staticmethod
    def forward(ctx, x, weight, layer, hw, stats, add_slot, sink_slot, producer, wlink=None):
        rt = _need_rt(layer)
        N = x.shape[0]
        pl = layer._plan(N, hw[0], hw[1])
        y = torch.empty(N, pl.fd.OP, pl.fd.OQ, layer.geom.cout, dtype=x.dtype, device=x.device)
        ops.conv_igemm(pl.fd, x, rt.w_fwd, y, stats=stats[0] if stats is not None else None)
        if any(ctx.needs_input_grad):
            rt.arena.expect_grad(rt.indices)
        ctx.save_for_backward(x)
        ctx.layer, ctx.pl, ctx.hw = layer, pl, hw
        ctx.add_slot, ctx.sink_slot, ctx.producer = add_slot, sink_slot, producer
        # latched: forward and backward of one step agree about the side stream even if the allocator-pressure
        # back-off of streams.enabled() flips in between
        ctx.side = streams.enabled(x)
        ctx.wlink = None
        if wlink is not None and ctx.side and any(ctx.needs_input_grad):
            # the producer of dy may launch the weight gradient in pieces of the batch (WgradLink)
            ctx.wlink = wlink
            state = {}

            def piece(dy_all, n0, n1, tmp):
                _stem_wgrad(layer, layer._plan(n1 - n0, hw[0], hw[1]), x[n0:n1], dy_all[n0:n1], tmp)

            def launch(dy_all, n0, n1, i, parts):
                dev = dy_all.device
                on_main = parts > 1 and config.stem_wgrad_main_last()      # the last piece
                if on_main and i == parts - 1:
                    # the last piece on THIS stream, into a buffer of its own; dW += it once the side stream's pieces
                    # are in (same order every step)
                    tmp = ops.zeros(layer.geom.cout, P.STEM_K * P.STEM_ROW, dtype=torch.float32, device=dev)
                    piece(dy_all, n0, n1, tmp)
                    streams.wait_stream(torch.cuda.current_stream(dev), streams.side_stream(dev))
                    _stem_wgrad_finish(layer, tmp)
                    return
                with streams.on_side(dev, reads=(x, dy_all), in_backward=True):
                    if 'tmp' not in state:
                        state['tmp'] = ops.zeros(layer.geom.cout, P.STEM_K * P.STEM_ROW, dtype=torch.float32, device=dev)
                    piece(dy_all, n0, n1, state['tmp'])
                    if i == parts - (2 if on_main else 1):
                        _stem_wgrad_finish(layer, state.pop('tmp'))
            wlink.launch = launch
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        layer, pl = ctx.layer, ctx.pl
        rt = layer._rt
        g = layer.geom
        streams.autograd_node_entry(dy.device)
        dy = dy.contiguous()
        dx = None
        if ctx.needs_input_grad[0]:
            N = x.shape[0]
            alloc = ops.zeros if pl.dgrad_zero else torch.empty
            dx = alloc(N, ctx.hw[0], ctx.hw[1], g.cin, dtype=dy.dtype, device=dy.device)
            extra = ctx.add_slot.take() if ctx.add_slot is not None else None
            if extra is not None and (len(pl.dds) != 1 or pl.dgrad_zero):
                raise RuntimeError('residual-gradient fusion needs a single dense data-gradient conv')
            link = ctx.producer
            fuse = (link is not None and dy.dtype == torch.bfloat16 and not pl.dgrad_zero and
                    ctx.sink_slot is None and config.fused_bn_backward())
            slab, slab2, off = None, None, 0
            if fuse:
                tiles = sum(ops.conv_tiles(d) for d in pl.dds)
                slab = torch.empty(ops.bn_partial_floats(tiles, g.cin, False), dtype=torch.float32,
                                   device=dy.device)
                # the downsample branch's BatchNorm behind the same gradient: reduced by this launch too (the
                # instantiation covers dense 1x1 launches of at least 128 columns: conv1 of the block that follows)
                rl = link.res_link
                if (rl is not None and rl.relu_mode == 0 and rl.y is not None and config.fused_bn_backward2() and
                        len(pl.dds) == 1 and g.k == 1 and g.stride == 1 and g.pad == 0 and g.cout % 64 == 0 and
                        g.cin >= 128):
                    slab2 = torch.empty(ops.bn_partial_floats(tiles, g.cin, False), dtype=torch.float32,
                                        device=dy.device)
            for d in pl.dds:
                bnb = None
                if fuse:
                    bnb = dict(y=link.y, mask=link.mask, mean=link.st[0], invstd=link.st[1],
                               scale=link.st[2], shift=link.st[3], relu=link.relu_mode, partial=slab,
                               tile_off=off)
                    if slab2 is not None:
                        bnb.update(y2=link.res_link.y, mean2=link.res_link.st[0], invstd2=link.res_link.st[1],
                                   partial2=slab2)
                    off += ops.conv_tiles(d)
                ops.conv_igemm(d, dy, rt.w_dgrad[id(d.pack)], dx, residual=extra, bnb=bnb)
                if slab2 is not None and not bnb.get('partial2_done', False):
                    slab2 = None          # the library took the launch without the second layer (ops.conv_igemm)
            if fuse:
                link.fused = (slab, off)
                if slab2 is not None:
                    link.res_link.fused = (slab2, off)
            if ctx.sink_slot is not None:
                ctx.sink_slot.put(dx)
                dx = None
        def wgrad():
            if layer.is_stem:
                tmp = ops.zeros(g.cout, P.STEM_K * P.STEM_ROW, dtype=torch.float32, device=dy.device)
                _stem_wgrad(layer, pl, x, dy, tmp)
                _stem_wgrad_finish(layer, tmp)
            else:
                ops.conv_wgrad(pl.wd, x, dy.view(-1, g.cout), rt.dw)
        wl = ctx.wlink
        if wl is not None:
            wl.launch = None                      # (the closure holds x)
        if wl is not None and wl.done:
            pass                                  # launched by the producer of dy, piece by piece
        elif ctx.side:
            # dW feeds only the optimizer: off the dgrad -> BatchNorm-backward chain (hip/streams.py)
            rows = config.side_urgent_rows()
            streams.side_later(dy.device, wgrad, reads=(x, dy), urgent=rows > 0 and dy.numel() // g.cout >= rows)
        else:
            wgrad()
        rt.arena.grad_ready(rt.indices)
        return dx, None, None, None, None, None, None, None, None


class Conv2D(Layer):
    """Bias-free 2-D convolution, NHWC activations, weight logically [Cout,Cin,kh,kw]."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias_attr=False, **_):
        super().__init__()
        assert bias_attr is False, 'only bias-free convs are on the hot path'
        assert dilation == 1 and groups == 1
        self.geom = P.ConvGeom(in_channels, out_channels, kernel_size, stride, padding)
        self.is_stem = (in_channels, kernel_size, stride, padding) == (3, 7, 2, 3)
        self.weight = tnn.Parameter(torch.empty(out_channels, in_channels, kernel_size, kernel_size,
                                                device=config.get_device()))
        self._rt = None
        self._plans = {}

    def _plan(self, N, H, W):
        key = (N, H, W)
        pl = self._plans.get(key)
        if pl is None:
            rt = _need_rt(self)
            g = self.geom
            if self.is_stem:
                fd = P.stem_desc(g.cout, N, H, W)
                pl = SimpleNamespace(fd=fd, wd=fd, dds=[], dgrad_zero=False)
            else:
                dds, skipped = ([], False)
                if rt.dgrad_packs is not None:
                    dds, skipped = P.dgrad_plan(g, N, H, W, packs=rt.dgrad_packs)
                pl = SimpleNamespace(fd=P.fwd_desc(g, N, H, W), wd=P.wgrad_desc(g, N, H, W),
                                     dds=dds, dgrad_zero=skipped)
            self._plans[key] = pl
        return pl

    def forward(self, x, hw=None, want_stats=False, add_slot=None, sink_slot=None, producer=None):
        """x: NHWC compute-dtype tensor (the stem takes the zero-padded image + hw=(H, W)).
        want_stats: also return the fused BatchNorm statistics slab written by the conv epilogue
        ((slab, tiles); None when the dtype has no fused path) -> (y, stats).
        add_slot / sink_slot: residual-fork gradient hand-off (GradSlot): this conv's backward adds
        the slot's tensor to dx in the kernel epilogue / deposits its dx there instead of
        returning it.
        producer: BNLink of the BatchNorm layer whose output is x, given ONLY when this conv's
        data-gradient launch produces the complete gradient of x (sole consumer, or the residual fork
        folded in through add_slot): the launch then also does that BatchNorm's backward reduction."""
        if hw is None:
            hw = (x.shape[1], x.shape[2])
        stats = None
        if want_stats and x.dtype == torch.bfloat16 and config.fused_bn_stats():
            stats = ops.conv_stats_buffer(self._plan(x.shape[0], hw[0], hw[1]).fd, x.device)
        wlink = WgradLink() if (self.is_stem and config.stem_wgrad_parts() > 1) else None
        y = _ConvFn.apply(x, self.weight, self, hw, stats, add_slot, sink_slot, producer, wlink)
        if wlink is not None and wlink.launch is not None:
            y._passl_wgrad_link = wlink
        return (y, stats) if want_stats else y

    @torch.no_grad()
    def infer(self, x, bn=None, residual=None, relu=False, hw=None):
        """conv + (BatchNorm with running stats) + residual + ReLU in ONE kernel (epilogue)."""
        rt = _need_rt(self)
        if hw is None:
            hw = (x.shape[1], x.shape[2])
        N = x.shape[0]
        pl = self._plan(N, hw[0], hw[1])
        y = torch.empty(N, pl.fd.OP, pl.fd.OQ, self.geom.cout, dtype=x.dtype, device=x.device)
        scale = shift = None
        if bn is not None:
            scale, shift = bn.infer_affine()
        ops.conv_igemm(pl.fd, x, rt.w_fwd, y, scale=scale, shift=shift, residual=residual, relu=relu)
        return y


# =============================================================================== batch norm
def _collectives_active():
    from ..core.sync_utils import collectives_active
    return collectives_active()


def convert_sync_batchnorm(module):
    """paddle.nn.SyncBatchNorm.convert_sync_batchnorm: every BatchNorm of `module` computes its training statistics
    over the batches of ALL data-parallel ranks from now on (no-op without a process group)."""
    for m in module.modules():
        if isinstance(m, _BatchNormBase):
            m._sync = True
    return module


class _BNActFn(Function):
    @staticmethod
    def forward(ctx, y, gamma, beta, residual, layer, relu, partial, res_slot, link_box, res_link=None):
        has_res = residual is not None
        # SyncBatchNorm (convert_sync_batchnorm): statistics over every rank's batch when a process group is up
        ctx.sync = bool(getattr(layer, '_sync', False)) and _collectives_active()
        z, st, mask = ops.bn_train_fwd(y, gamma.detach(), beta.detach(), layer._mean,
                                       layer._variance, residual, relu, layer._momentum,
                                       layer._epsilon, partial=partial,
                                       want_mask=relu and has_res, sync=ctx.sync)
        # ReLU mask for the backward: recomputed from y (no residual) or the bit mask (residual):
        # the output z is never re-read by this layer's backward.
        ctx.relu_mode = 0 if not relu else (3 if has_res else 2)
        if layer._rt is not None and any(ctx.needs_input_grad):
            layer._rt.arena.expect_grad(layer._rt.indices)
        ctx.save_for_backward(y, st, mask)
        ctx.layer, ctx.has_res, ctx.res_slot = layer, has_res, res_slot
        ctx.link = None
        if link_box is not None:
            ctx.link = link_box[0] = BNLink(y, st, mask, ctx.relu_mode,
                                            res_link if (has_res and ctx.relu_mode == 3 and res_slot is None) else None)
        return z

    @staticmethod
    def backward(ctx, dz):
        y, st, mask = ctx.saved_tensors
        layer = ctx.layer
        streams.autograd_node_entry(dz.device)
        if layer.affine:
            for p in (layer.weight, layer.bias):
                if p.grad is None:
                    p.grad = ops.zeros_like(p)
            gamma, dgamma, dbeta = layer.weight.detach(), layer.weight.grad, layer.bias.grad
        else:
            gamma, dgamma, dbeta = layer._const[0], layer._dscratch[0], layer._dscratch[1]
        want_dres = ctx.has_res and (ctx.needs_input_grad[3] or ctx.res_slot is not None)
        # the consumer conv's data-gradient launch may already have masked dz and reduced it
        fused = ctx.link.fused if ctx.link is not None else None
        if ctx.link is not None:
            ctx.link.y = ctx.link.st = ctx.link.mask = ctx.link.fused = ctx.link.res_link = None   # drop the references
        if fused is not None and fused[0].is_cuda:
            # the slab may have been written by a launch on ANOTHER stream (a downsample branch's layer runs its
            # backward on the side stream; its slab comes from the main stream's data-gradient launch and pool)
            fused[0].record_stream(torch.cuda.current_stream(dz.device))
        dx, dres = ops.bn_bwd(dz.contiguous(), mask, y, gamma, st[0], st[1],
                              dgamma, dbeta, relu=ctx.relu_mode,
                              want_dres=want_dres, scale=st[2], shift=st[3], fused=fused, sync=ctx.sync)
        if ctx.res_slot is not None:
            ctx.res_slot.put(dres)
            dres = None
        if layer._rt is not None:
            layer._rt.arena.grad_ready(layer._rt.indices)
        return dx, None, None, dres, None, None, None, None, None, None


class _BatchNormBase(Layer):
    """BatchNorm over the channel (last) axis of NHWC rows.  ``_mean`` / ``_variance`` keep the
    reference's state_dict key names; ``_use_global_stats`` is what freeze_batchnorm_statictis
    (passl_v110/modules/freeze.py:18-23) flips on the key encoder."""

    def __init__(self, num_features, momentum=0.9, epsilon=1e-05, weight_attr=None,
                 bias_attr=None, data_format='NCHW', use_global_stats=None, name=None):
        super().__init__()
        dev = config.get_device()
        self._momentum, self._epsilon = momentum, epsilon
        self._use_global_stats = use_global_stats
        self.num_features = num_features
        # weight_attr=False / bias_attr=False (both or neither): no gamma / beta (the last BatchNorm of MoCo-v3's
        # projector and predictor, passl/models/mocov3.py:152-157).  The kernels read constant 1 / 0 vectors and
        # write the unused d-gamma / d-beta into a scratch pair; neither shows up in parameters() or state_dict().
        if (weight_attr is False) != (bias_attr is False):
            raise NotImplementedError('BatchNorm with only one of weight / bias')
        self.affine = weight_attr is not False
        if self.affine:
            self.weight = tnn.Parameter(torch.ones(num_features, device=dev))
            self.bias = tnn.Parameter(torch.zeros(num_features, device=dev))
        else:
            self.weight = self.bias = None
            self.register_buffer('_const', torch.stack([torch.ones(num_features, device=dev),
                                                        torch.zeros(num_features, device=dev)]), persistent=False)
            self.register_buffer('_dscratch', torch.zeros(2, num_features, device=dev), persistent=False)
        self.register_buffer('_mean', torch.zeros(num_features, device=dev))
        self.register_buffer('_variance', torch.ones(num_features, device=dev))
        self._rt = None

    def uses_global_stats(self):
        return bool(self._use_global_stats) if self._use_global_stats is not None \
            else (not self.training)

    def infer_affine(self):
        """(scale, shift) of y = x*scale + shift with the running statistics."""
        rt = self._rt
        if rt is not None and rt.arena.bn_affine is not None:
            s, e = rt.bn_slice
            return rt.arena.bn_affine[0][s:e], rt.arena.bn_affine[1][s:e]
        gamma, beta = self.gamma_beta()
        scale = gamma * torch.rsqrt(self._variance + self._epsilon)
        return scale, beta - self._mean * scale

    def gamma_beta(self):
        if self.affine:
            return self.weight.detach(), self.bias.detach()
        return self._const[0], self._const[1]

    def forward(self, y, residual=None, relu=False, stats=None, res_slot=None, res_link=None):
        """stats: fused statistics from the producing conv's epilogue (Conv2D.forward(...,
        want_stats=True)); res_slot: GradSlot that receives the residual branch's gradient; res_link: bn_link of the
        residual when it is the output of a BatchNorm that nothing else consumes (BNLink.res_link)."""
        if self.uses_global_stats():
            if torch.is_grad_enabled() and (y.requires_grad or (self.affine and self.weight.requires_grad)):
                raise NotImplementedError('frozen BatchNorm inside a differentiated graph is not on '
                                          'the MoCo hot path (key encoder runs under no_grad)')
            scale, shift = self.infer_affine()
            return ops.bn_apply(y, scale, shift, residual, relu)
        # the output carries a BNLink so that a sole-consumer conv can take over the backward reduction
        box = [None] if (torch.is_grad_enabled() and y.dtype == torch.bfloat16 and
                         config.fused_bn_backward()) else None
        gamma, beta = (self.weight, self.bias) if self.affine else (self._const[0], self._const[1])
        z = _BNActFn.apply(y, gamma, beta, residual, self, relu, stats, res_slot, box, res_link)
        if box is not None and box[0] is not None:
            z._passl_bn_link = box[0]
        return z


class BatchNorm2D(_BatchNormBase):
    pass


class BatchNorm1D(_BatchNormBase):
    pass


# =============================================================================== pooling
class _MaxPoolFn(Function):
    @staticmethod
    def forward(ctx, x):
        y, idx = ops.maxpool_fwd(x)
        ctx.save_for_backward(idx)
        ctx.hw = (x.shape[1], x.shape[2])
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        return ops.maxpool_bwd(dy.contiguous(), idx, ctx.hw[0], ctx.hw[1])


class _BNReluMaxPoolFn(Function):
    """Training-mode BatchNorm + ReLU + MaxPool2D(3, 2, 1) in one pass per direction (csrc/stem_pool.hip): the stem of
    the trainable trunk (resnetimagenet.py:196-198).  Saves the BatchNorm input, the batch statistics and one index
    byte per pooled element; neither the BatchNorm output nor the pool's input gradient is ever written."""

    @staticmethod
    def forward(ctx, y, gamma, beta, layer, partial, wlink=None):
        _, st, _ = ops.bn_train_fwd(y, gamma.detach(), beta.detach(), layer._mean, layer._variance, None, True,
                                    layer._momentum, layer._epsilon, partial=partial, apply=False)
        out, idx = ops.bn_relu_maxpool_fwd(y, st)
        if layer._rt is not None and any(ctx.needs_input_grad):
            layer._rt.arena.expect_grad(layer._rt.indices)
        ctx.save_for_backward(y, st, idx)
        ctx.layer, ctx.wlink = layer, wlink
        return out

    @staticmethod
    def backward(ctx, dout):
        y, st, idx = ctx.saved_tensors
        layer, wl = ctx.layer, ctx.wlink
        streams.autograd_node_entry(dout.device)
        # These launches and the stem's weight gradient are the tail of the backward pass: weight gradients still queued
        # for the side stream go there now, next to them, not behind them.
        if config.stem_tail_flush():
            streams.flush_side(dout.device)
        for p in (layer.weight, layer.bias):
            if p.grad is None:
                p.grad = ops.zeros_like(p)
        piecewise = wl is not None and wl.launch is not None and not wl.done
        dx = ops.bn_relu_maxpool_bwd(dout.contiguous(), idx, y, layer.weight.detach(), st, layer.weight.grad,
                                     layer.bias.grad, parts=config.stem_wgrad_parts() if piecewise else 1,
                                     on_part=wl.launch if piecewise else None)
        if piecewise:
            wl.done = True
        if layer._rt is not None:
            layer._rt.arena.grad_ready(layer._rt.indices)
        return dx, None, None, None, None, None


def bn_relu_maxpool(bn, y, stats=None):
    """bn(y, relu=True) followed by MaxPool2D(3, 2, 1).  One fused pass per direction when `bn` computes batch
    statistics of this rank (training mode, affine, no SyncBatchNorm) and the shape is covered; the two layers
    otherwise."""
    fused = (config.fused_stem_pool() and torch.is_grad_enabled() and not bn.uses_global_stats() and bn.affine and
             not (getattr(bn, '_sync', False) and _collectives_active()) and y.is_cuda and
             ops.bn_relu_maxpool_supported(y))
    if not fused:
        return _MaxPoolFn.apply(bn(y, relu=True, stats=stats))
    return _BNReluMaxPoolFn.apply(y, bn.weight, bn.bias, bn, stats, wgrad_link(y))


class MaxPool2D(Layer):
    def __init__(self, kernel_size=3, stride=2, padding=1):
        super().__init__()
        assert (kernel_size, stride, padding) == (3, 2, 1), 'only the ResNet stem pool is built'

    def forward(self, x):
        return _MaxPoolFn.apply(x)


class _AvgPoolFn(Function):
    @staticmethod
    def forward(ctx, x):
        ctx.hw = (x.shape[1], x.shape[2])
        return ops.avgpool_fwd(x)

    @staticmethod
    def backward(ctx, dy):
        return ops.avgpool_bwd(dy.contiguous(), ctx.hw[0], ctx.hw[1])


class AdaptiveAvgPool2D(Layer):
    def __init__(self, output_size=(1, 1)):
        super().__init__()
        assert tuple(output_size) == (1, 1) if not isinstance(output_size, int) else output_size == 1

    def forward(self, x):
        """[N,H,W,C] -> [N,C]"""
        return _AvgPoolFn.apply(x)


class ReLU(Layer):
    """Marker layer: ReLU is always fused into the producing kernel's epilogue
    (BatchNorm-apply, conv epilogue or Linear epilogue)."""

    def forward(self, x):
        raise RuntimeError('ReLU is fused into the preceding layer on the HIP path')


# =============================================================================== linear
class _LinearFn(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, layer, relu, out_f32, residual):
        rt = _need_rt(layer)
        N = x.shape[0]
        pl = layer._plan(N)
        y = torch.empty(N, layer.out_features, dtype=torch.float32 if out_f32 else x.dtype,
                        device=x.device)
        ops.conv_igemm(pl.fd, x, rt.w_fwd, y, shift=bias.detach() if bias is not None else None,
                       relu=relu, out_f32=out_f32, residual=residual)
        if any(ctx.needs_input_grad):
            rt.arena.expect_grad(rt.indices)
        ctx.save_for_backward(x, y if relu else None)
        ctx.layer, ctx.pl, ctx.relu = layer, pl, relu
        ctx.has_res = residual is not None
        ctx.side = streams.enabled(x)              # latched for the backward (see _ConvFn)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y = ctx.saved_tensors
        layer, pl = ctx.layer, ctx.pl
        rt = layer._rt
        dy = dy.contiguous()
        if dy.dtype != x.dtype:
            dy = ops.cast_to(dy, x.dtype)
        if ctx.relu:
            dy = ops.relu_bwd(dy, y)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(x.shape[0], layer.in_features, dtype=x.dtype, device=x.device)
            d = pl.dds[0]
            ops.conv_igemm(d, dy, rt.w_dgrad[id(d.pack)], dx)
        want_bias = layer.bias is not None and layer.bias.requires_grad
        if ctx.side and not config.side_reductions():
            streams.side_later(dy.device, lambda: ops.conv_wgrad(pl.wd, x, dy, rt.dw), reads=(x, dy))
            if want_bias:
                ops.colsum_into(dy, layer.bias.grad, accumulate=True)
        elif ctx.side:
            # the bias gradient (column sums of dy) rides along: like dW it feeds only the optimizer, and its two small
            # launches are latency-bound — off the data-gradient chain (MAE: 83 of them per step, 1.1 ms)
            bias_grad = layer.bias.grad if want_bias else None

            def side_work():
                ops.conv_wgrad(pl.wd, x, dy, rt.dw)
                if want_bias:
                    ops.colsum_into(dy, bias_grad, accumulate=True)     # straight into the arena's gradient
            streams.side_later(dy.device, side_work, reads=(x, dy))
        else:
            ops.conv_wgrad(pl.wd, x, dy, rt.dw)
            if want_bias:
                ops.colsum_into(dy, layer.bias.grad, accumulate=True)
        rt.arena.grad_ready(rt.indices)
        # y = x W + b + residual: the residual branch's gradient is dy itself
        return dx, None, None, None, None, None, (dy if ctx.has_res else None)


class Linear(Layer):
    """y = x W + b with W logically [in, out] (the reference's paddle.nn.Linear layout)."""

    def __init__(self, in_features, out_features, weight_attr=None, bias_attr=None, name=None):
        super().__init__()
        dev = config.get_device()
        self.in_features, self.out_features = in_features, out_features
        self.geom = P.ConvGeom(in_features, out_features, 1, 1, 0)
        self.weight = tnn.Parameter(torch.empty(in_features, out_features, device=dev))
        self.bias = None if bias_attr is False else tnn.Parameter(torch.zeros(out_features, device=dev))
        self._rt = None
        self._plans = {}

    def _plan(self, N):
        pl = self._plans.get(N)
        if pl is None:
            rt = _need_rt(self)
            dds = []
            if rt.dgrad_packs is not None:
                dds, _ = P.dgrad_plan(self.geom, N, 1, 1, packs=rt.dgrad_packs)
            pl = SimpleNamespace(fd=P.fwd_desc(self.geom, N, 1, 1), wd=P.wgrad_desc(self.geom, N, 1, 1),
                                 dds=dds)
            self._plans[N] = pl
        return pl

    def forward(self, x, relu=False, out_f32=False, residual=None):
        """residual: tensor shaped like the output, added in the GEMM epilogue (before the ReLU)."""
        return _LinearFn.apply(x, self.weight, self.bias, self, relu, out_f32, residual)


class _ToComputeFn(Function):
    """fp32 rows -> compute dtype (bf16) with a HIP cast kernel; gradient comes back as fp32."""

    @staticmethod
    def forward(ctx, x):
        y = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
        ops.cast_bf16(x.contiguous().view(-1), y.view(-1))
        return y

    @staticmethod
    def backward(ctx, dy):
        return ops.cast_f32(dy)


def to_compute(x, dtype):
    """Cast an fp32 activation to the compute dtype (identity for fp32 compute)."""
    if dtype == torch.float32 or x.dtype == dtype:
        return x
    return _ToComputeFn.apply(x)


# =============================================================================== ViT pieces
class _LayerNormFn(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, layer):
        y, mean, rstd = ops.layernorm_fwd(x.contiguous(), gamma.detach(), beta.detach(), layer._epsilon)
        if layer._rt is not None and any(ctx.needs_input_grad):
            layer._rt.arena.expect_grad(layer._rt.indices)
        ctx.save_for_backward(x, mean, rstd)
        ctx.layer = layer
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, rstd = ctx.saved_tensors
        layer = ctx.layer
        for p in (layer.weight, layer.bias):
            if p.grad is None:
                p.grad = torch.zeros_like(p)
        dx = _ln_backward(dy.contiguous(), x, layer, mean, rstd, None)
        if layer._rt is not None:
            layer._rt.arena.grad_ready(layer._rt.indices)
        return dx, None, None, None


def _ln_backward(dy, x, layer, mean, rstd, dres):
    """LayerNorm backward: the input gradient on the current stream; the fold of the per-block d-gamma / d-beta
    partials — a latency-bound launch nothing in the backward chain waits for (only the optimizer / the gradient
    reducer read its result, and both join the side stream first) — on the side stream when there is one."""
    if config.side_reductions() and streams.enabled(dy):
        dx, partials, fold = ops.layernorm_bwd(dy, x, layer.weight.detach(), mean, rstd, layer.weight.grad,
                                               layer.bias.grad, dres=dres, defer_params=True)
        streams.side_later(dy.device, fold, reads=(partials,))
        return dx
    return ops.layernorm_bwd(dy, x, layer.weight.detach(), mean, rstd, layer.weight.grad, layer.bias.grad, dres=dres)


class _LayerNormForkFn(Function):
    """(LN(x), x): the pre-norm residual fork `x + f(LN(x))`.  The second output is x itself, handed to
    the residual consumer (the GEMM epilogue of f's last Linear); in backward the gradient that comes
    back through it is added to the LayerNorm gradient INSIDE the LayerNorm backward kernel, so the fork
    costs no separate elementwise add."""

    @staticmethod
    def forward(ctx, x, gamma, beta, layer):
        x = x.contiguous()
        y, mean, rstd = ops.layernorm_fwd(x, gamma.detach(), beta.detach(), layer._epsilon)
        if layer._rt is not None and any(ctx.needs_input_grad):
            layer._rt.arena.expect_grad(layer._rt.indices)
        ctx.save_for_backward(x, mean, rstd)
        ctx.layer = layer
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dres):
        x, mean, rstd = ctx.saved_tensors
        layer = ctx.layer
        for p in (layer.weight, layer.bias):
            if p.grad is None:
                p.grad = torch.zeros_like(p)
        if dres is not None:
            dres = dres.contiguous()
            if dres.dtype != x.dtype:
                dres = dres.to(x.dtype)
        dx = _ln_backward(dy.contiguous(), x, layer, mean, rstd, dres)
        if layer._rt is not None:
            layer._rt.arena.grad_ready(layer._rt.indices)
        return dx, None, None, None


class LayerNorm(Layer):
    """paddle.nn.LayerNorm over the last axis (biased variance), rows [M, C] in the compute dtype."""

    def fork(self, x):
        """-> (LN(x), x_for_the_residual_add); see _LayerNormForkFn."""
        return _LayerNormForkFn.apply(x, self.weight, self.bias, self)

    def __init__(self, normalized_shape, epsilon=1e-05, weight_attr=None, bias_attr=None, name=None):
        super().__init__()
        dev = config.get_device()
        n = normalized_shape if isinstance(normalized_shape, int) else normalized_shape[-1]
        self._epsilon = epsilon
        self.weight = tnn.Parameter(torch.ones(n, device=dev))
        self.bias = tnn.Parameter(torch.zeros(n, device=dev))
        self._rt = None

    def forward(self, x):
        return _LayerNormFn.apply(x, self.weight, self.bias, self)


class _GeluFn(Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return ops.gelu_fwd(x.contiguous())

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return ops.gelu_bwd(dy.contiguous(), x)


def gelu(x):
    return _GeluFn.apply(x)


class GELU(Layer):
    def forward(self, x):
        return gelu(x)


class _TanhFn(Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return ops.tanh_fwd(x.contiguous())

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return ops.tanh_bwd(dy.contiguous(), x)


class Tanh(Layer):
    def forward(self, x):
        return _TanhFn.apply(x)


class _AttentionFn(Function):
    """softmax(q k^T * scale) v per (image, head) on the fused qkv projection [B*T, 3*H*d]."""

    @staticmethod
    def forward(ctx, qkv, B, T, H, DH, scale, causal):
        out, lse = ops.attention_fwd(qkv.contiguous(), B, T, H, DH, scale, causal)
        ctx.save_for_backward(qkv, out, lse)
        ctx.dims = (B, T, H, DH, scale, causal)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, out, lse = ctx.saved_tensors
        B, T, H, DH, scale, causal = ctx.dims
        return ops.attention_bwd(qkv, out, dout.contiguous(), lse, B, T, H, DH, scale, causal), None, None, \
            None, None, None, None


def param_expect_grad(*params):
    """Called from the FORWARD of a custom Function that will report ``param_grad_ready`` for these raw arena parameters
    from its backward: a parameter used by several forward nodes of one step (a trunk run once per view) is complete —
    ready for its all-reduce bucket — only after the last contribution (EncoderArena.expect_grad)."""
    if not torch.is_grad_enabled():
        return
    for p in params:
        a = getattr(p, '_passl_arena', None)
        if a is not None and p.requires_grad:
            a.expect_grad([p._passl_index])


def param_grad_ready(*params):
    """Report raw arena parameters as reduced-ready from a custom backward (no-op outside an arena)."""
    for p in params:
        a = getattr(p, '_passl_arena', None)
        if a is not None:
            a.param_grad_ready(p)


def attention(qkv, B, T, H, DH, scale, causal=False):
    return _AttentionFn.apply(qkv, B, T, H, DH, float(scale), bool(causal))


class _QuickGeluFn(Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return ops.quick_gelu_fwd(x.contiguous())

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return ops.quick_gelu_bwd(dy.contiguous(), x)


class QuickGELU(Layer):
    """x * sigmoid(1.702 x) (passl_v110/modeling/backbones/base_transformer.py:25-28)."""

    def forward(self, x):
        return _QuickGeluFn.apply(x)


class _GatherRowsFn(Function):
    """out[r] = x[idx[r]] for distinct row indices (class token / EOT token selection)."""

    @staticmethod
    def forward(ctx, x, idx):
        ctx.save_for_backward(idx)
        ctx.rows = x.shape[0]
        return ops.gather_rows(x.contiguous(), idx)

    @staticmethod
    def backward(ctx, dout):
        (idx,) = ctx.saved_tensors
        return ops.scatter_rows(dout.contiguous(), idx, ctx.rows), None


def gather_rows(x, idx):
    return _GatherRowsFn.apply(x, idx)


# =============================================================================== head pieces
class _L2NormFn(Function):
    @staticmethod
    def forward(ctx, x, eps):
        y, norm = ops.l2norm_fwd(x.contiguous(), eps)
        ctx.save_for_backward(y, norm)
        return y

    @staticmethod
    def backward(ctx, dy):
        y, norm = ctx.saved_tensors
        return ops.l2norm_bwd(dy.contiguous(), y, norm, torch.float32), None


def normalize(x, axis=1, epsilon=1e-12):
    """paddle.nn.functional.normalize(x, axis=1) for [N, D] fp32 rows."""
    assert x.dim() == 2 and axis in (1, -1)
    return _L2NormFn.apply(x, epsilon)


def infonce(q, k, queue, T):
    """Fused InfoNCE — lives in passl_amd/loss/moco.py (``passl.loss.moco``)."""
    from ..loss.moco import info_nce
    return info_nce(q, k, queue, T)


# =============================================================================== arena
class EncoderArena:
    """Flat fp32 storage for one encoder (see module docstring)."""

    ALIGN = 8   # elements; keeps every slot 32-byte aligned in fp32 and 16-byte aligned in bf16

    def __init__(self, module, trainable=True, dtype=None, exclude=(), exclude_params=()):
        """exclude: sub-layers whose own parameters / statistics live elsewhere (a frozen layer inside a trainable
        encoder gets a non-trainable arena of its own, e.g. MoCo-v3's patch embedding).  exclude_params: single
        parameters that stay ordinary tensors outside the arena (SimSiam's projector bias whose gradient is
        switched off, passl/models/simsiam.py:61)."""
        self.module = module
        skip = {id(m) for m in exclude}
        skip_p = {id(q) for q in exclude_params}
        self.trainable = trainable
        self.dtype = dtype or config.get_compute_dtype()
        self.reducer = None
        self._uses = {}          # parameter index -> forward uses of this step still waiting for their backward
        self.bn_affine = None
        params = []      # (owner, name, kind)
        seen = set()
        for mod in module.modules():
            if id(mod) in skip:
                continue
            for name, p in mod._parameters.items():
                if p is None or id(p) in seen or id(p) in skip_p:
                    continue
                seen.add(id(p))
                kind = 'conv' if ((isinstance(mod, Conv2D) or getattr(mod, 'krsc_weight', False))
                                  and name == 'weight') else \
                    ('linw' if isinstance(mod, Linear) and name == 'weight' else 'vec')
                params.append((mod, name, kind))
        stats = [(mod, n) for mod in module.modules() if isinstance(mod, _BatchNormBase) and id(mod) not in skip
                 for n in ('_mean', '_variance')]
        if not params:
            raise ValueError('EncoderArena over a module without parameters')
        if trainable and any(not m._parameters[n].requires_grad for m, n, _k in params):
            raise NotImplementedError(
                'a trainable EncoderArena holds only trainable parameters: frozen sub-layers (e.g. '
                'ResNet(frozen_stages >= 0)) belong in their own non-trainable arena — see '
                'modeling/architectures/clas.py; the pre-training architectures use frozen_stages -1')
        dev = params[0][0]._parameters[params[0][1]].device
        self.device = dev

        def aligned(n):
            return (n + self.ALIGN - 1) // self.ALIGN * self.ALIGN

        off = 0
        self.param_slices = []   # (off, numel) per trainable parameter, registration order
        slots = []
        for mod, name, kind in params:
            n = mod._parameters[name].numel()
            slots.append((mod, name, kind, off, n))
            self.param_slices.append((off, n))
            off += aligned(n)
        self.n_train = off
        stat_slots = []
        for mod, name in stats:
            n = mod._buffers[name].numel()
            stat_slots.append((mod, name, off, n))
            off += aligned(n)
        self.total = off
        self.flat = torch.zeros(self.total, dtype=torch.float32, device=dev)
        self.grads = torch.zeros(self.n_train, dtype=torch.float32, device=dev) if trainable else None
        self.lp = torch.zeros(self.total, dtype=torch.bfloat16, device=dev) \
            if self.dtype == torch.bfloat16 else None
        self.packer = WeightPacker()
        owners = {}
        for index, (mod, name, kind, o, n) in enumerate(slots):
            old = mod._parameters[name].detach()
            seg = self.flat[o:o + n]
            if kind == 'conv':
                K, Cc, R, S = old.shape
                seg.view(K, R, S, Cc).copy_(old.permute(0, 2, 3, 1))
                view = seg.view(K, R, S, Cc).permute(0, 3, 1, 2)
                gview = self.grads[o:o + n].view(K, R, S, Cc).permute(0, 3, 1, 2) if trainable else None
            elif kind == 'linw':
                i, oo = old.shape
                seg.view(oo, i).copy_(old.t())
                view = seg.view(oo, i).t()
                gview = self.grads[o:o + n].view(oo, i).t() if trainable else None
            else:
                seg.copy_(old.reshape(-1))
                view = seg.view(old.shape)
                gview = self.grads[o:o + n].view(old.shape) if trainable else None
            p = tnn.Parameter(view, requires_grad=trainable)
            p._passl_arena = self
            p._passl_index = index          # position in param_slices (for param_grad_ready)
            if trainable:
                p.grad = gview
            mod._parameters[name] = p
            rt = owners.get(id(mod))
            if rt is None:
                rt = SimpleNamespace(arena=self, indices=[], dgrad_packs=None, w_dgrad={},
                                     bn_slice=None)
                owners[id(mod)] = rt
                mod._rt = rt
            rt.indices.append(index)
            if kind in ('conv', 'linw'):
                self._attach_gemm_layer(mod, rt, o, n)
        for mod, name, o, n in stat_slots:
            seg = self.flat[o:o + n]
            seg.copy_(mod._buffers[name])
            mod._buffers[name] = seg
        # index tensors for the one-shot inference-BN affine of all layers
        gi, bi, mi, vi, pos = [], [], [], [], 0
        slot_of = {(id(m), nm): (o, n) for m, nm, _k, o, n in slots}
        slot_of.update({(id(m), nm): (o, n) for m, nm, o, n in stat_slots})
        self._bn_eps = 1e-5
        for mod in module.modules():
            # (a BatchNorm without gamma / beta has no slot to fold from: it keeps the per-layer path)
            if isinstance(mod, _BatchNormBase) and mod.affine and id(mod) not in skip:
                Cc = mod.num_features
                for lst, nm in ((gi, 'weight'), (bi, 'bias'), (mi, '_mean'), (vi, '_variance')):
                    o, _ = slot_of[(id(mod), nm)]
                    lst.append(torch.arange(o, o + Cc))
                mod._rt.bn_slice = (pos, pos + Cc)
                pos += (Cc + 7) // 8 * 8          # keep every slice 32-byte aligned
                self._bn_eps = mod._epsilon
        self._bn_idx = None
        if gi:
            def cat_pad(lst):
                outs = []
                for t in lst:
                    padn = (-len(t)) % 8
                    outs.append(t)
                    if padn:
                        outs.append(t[:1].expand(padn))
                return torch.cat(outs).to(dev)
            self._bn_idx = tuple(cat_pad(lst) for lst in (gi, bi, mi, vi))
        self.packer.build(dev, self.dtype)
        # resolve packed-operand views now that the packer buffer exists
        for rt in owners.values():
            if getattr(rt, '_pending', None):
                rt._pending()
                rt._pending = None
        self._refresh_if_on_device()

    def _attach_gemm_layer(self, mod, rt, off, n):
        g = mod.geom
        rows = g.cout
        is_stem = getattr(mod, 'is_stem', False)
        rt.dw = self.grads[off:off + n].view(rows, -1) if self.trainable else None
        stem_pack = None
        if is_stem:
            stem_pack = P.stem_desc(g.cout, 1, 8, 8).pack
            self.packer.add(off, g.cout, g.k, g.k, g.cin, stem_pack)
        elif self.trainable and not getattr(mod, 'no_dgrad', False):
            rt.dgrad_packs = P.dgrad_packs(g)
            for pk in rt.dgrad_packs.values():
                self.packer.add(off, g.cout, g.k, g.k, g.cin, pk)

        def resolve():
            if is_stem:
                rt.w_fwd = self.packer.view(stem_pack, g.cout)
            elif self.lp is not None:
                rt.w_fwd = self.lp[off:off + n].view(rows, -1)
            else:
                rt.w_fwd = self.flat[off:off + n].view(rows, -1)
            if rt.dgrad_packs:
                rt.w_dgrad = {id(pk): self.packer.view(pk, g.cin) for pk in rt.dgrad_packs.values()}
        rt._pending = resolve

    # ---- per-step maintenance
    @torch.no_grad()
    def refresh(self):
        """Recompute the compute-dtype operand copies from the fp32 master weights."""
        if self.lp is not None:
            ops.cast_bf16(self.flat[:self.n_train], self.lp[:self.n_train])
        self.packer.run(self.flat)

    @torch.no_grad()
    def ema_from(self, other, m):
        """self = self*m + other*(1-m) over parameters AND BN statistics (one launch), then refresh
        the stem pack and the fused-BN affine.  moco.py:82-90 semantics (SURVEY §3.1 note A)."""
        ops.ema_update(self.flat, other.flat, m, self.lp)
        self.packer.run(self.flat)
        self.update_bn_affine()

    # ---- the same update in pieces (MoCo's key pipeline): parameters first, then the running statistics of one
    # group of BatchNorm layers at a time (each group's statistics and folded affine are contiguous: module order)
    @torch.no_grad()
    def ema_params_from(self, other, m):
        n = self.n_train
        ops.ema_update(self.flat[:n], other.flat[:n], m, self.lp[:n] if self.lp is not None else None)
        self.packer.run(self.flat)

    def bn_groups(self, groups):
        """groups: lists of sub-layers (in module order, together covering every BatchNorm of the arena) ->
        [(stat range in flat, affine range, index tensors)] for ema_stats_from."""
        out, pos = [], 0
        bns = [m for m in self.module.modules() if isinstance(m, _BatchNormBase) and m.affine]
        stat_start = {}
        off = self.n_train

        def aligned(n):
            return (n + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        for mod in [m for m in self.module.modules() if isinstance(m, _BatchNormBase)]:
            stat_start[id(mod)] = off
            off += 2 * aligned(mod.num_features)
        assert off == self.total
        done = 0
        for g in groups:
            mods = [m for sub in g for m in sub.modules() if isinstance(m, _BatchNormBase)]
            assert mods == bns[done:done + len(mods)], 'BatchNorm groups must follow the module order'
            s0 = stat_start[id(mods[0])]
            s1 = stat_start[id(mods[-1])] + 2 * aligned(mods[-1].num_features)
            a0 = mods[0]._rt.bn_slice[0]
            a1 = (mods[-1]._rt.bn_slice[1] + 7) // 8 * 8
            out.append(((s0, s1), (a0, a1), tuple(t[a0:a1].contiguous() for t in self._bn_idx)))
            done += len(mods)
        assert done == len(bns)
        return out

    @torch.no_grad()
    def ema_stats_from(self, other, m, group):
        (s0, s1), (a0, a1), idx = group
        ops.ema_update(self.flat[s0:s1], other.flat[s0:s1], m, None)
        if self.bn_affine is None:
            n = self._bn_idx[0].numel()
            self.bn_affine = (torch.empty(n, dtype=torch.float32, device=self.flat.device),
                              torch.empty(n, dtype=torch.float32, device=self.flat.device))
        ops.bn_fold(self.flat, idx, self._bn_eps, self.bn_affine[0][a0:a1], self.bn_affine[1][a0:a1])

    def _refresh_if_on_device(self):
        # Building a model on the host is allowed (config / registry / checkpoint tooling);
        # *running* it is not: refresh() and every layer raise on host tensors.
        if self.device.type == 'cuda':
            self.refresh()

    @torch.no_grad()
    def copy_from(self, other):
        self.flat.copy_(other.flat)
        self._refresh_if_on_device()
        self.update_bn_affine()

    @torch.no_grad()
    def update_bn_affine(self):
        if self._bn_idx is None:
            return
        n = self._bn_idx[0].numel()
        if self.bn_affine is None or self.bn_affine[0].numel() != n:
            self.bn_affine = (torch.empty(n, dtype=torch.float32, device=self.flat.device),
                              torch.empty(n, dtype=torch.float32, device=self.flat.device))
        if self.flat.is_cuda:
            ops.bn_fold(self.flat, self._bn_idx, self._bn_eps, *self.bn_affine)    # one launch, in place
        else:       # host-side model construction only (nothing runs there)
            g, b, m, v = (self.flat[i] for i in self._bn_idx)
            scale = g * torch.rsqrt(v + self._bn_eps)
            self.bn_affine[0].copy_(scale)
            self.bn_affine[1].copy_(b - m * scale)

    def clear_grad(self):
        if self.grads is not None:
            if self.grads.is_cuda:
                ops.fill_zero(self.grads)        # (a library launch: part of a recorded step plan)
            else:
                self.grads.zero_()

    def param_grad_ready(self, *params):
        """grad_ready for raw parameters (class / position embeddings, tokens, logit_scale ...) whose
        gradient is produced by one dedicated backward kernel: without the mark their bucket (and, since
        buckets launch in order, every later one) would only be reduced after the whole backward."""
        # through the same use counter as the layers' parameters: a Function whose forward ran twice in one step (two
        # views through one trunk) reports twice, and only the second report releases the bucket
        self.grad_ready([p._passl_index for p in params])

    def expect_grad(self, indices):
        """Called from the FORWARD of a layer (when autograd will run its backward) that will report ``grad_ready(indices)`` from its backward: a parameter
        used by several forward nodes of one step (SimSiam encodes its two views in two passes) receives one gradient
        contribution per use and is complete — ready for its all-reduce bucket — only after the last of them."""
        if self.reducer is not None:
            u = self._uses
            for i in indices:
                u[i] = u.get(i, 0) + 1

    def grad_ready(self, indices):
        """Called from the backward kernels' host code once the gradients of parameters
        `indices` (positions in param_slices) are enqueued on the compute stream."""
        if self.reducer is not None:
            u = self._uses
            for i in indices:
                left = u.get(i, 0) - 1
                if left > 0:
                    u[i] = left                # another forward use of this parameter has not run its backward yet
                    continue
                u.pop(i, None)
                self.reducer.mark_ready(i)
