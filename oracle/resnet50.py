"""Oracle: ResNet-50 trunk + MoCo-v2 projector, functional torch-CPU fp32.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows, line by line:

* topology     reference passl_v110/modeling/backbones/resnetimagenet.py:111-153
               (BottleneckBlock: 1x1 -> 3x3(stride) -> 1x1, stride on conv2,
               bias-free convs), :173-253 (stem 7x7 s2 p3, maxpool 3x3 s2 p1,
               layers 3/4/6/3, downsample 1x1(stride)+BN on the first block of
               a stage, num_classes=0 / with_pool=False returns the layer4 map)
* init         passl_v110/modeling/backbones/resnet.py:76-88 (conv: kaiming
               normal fan_out/relu; BN gamma=1 beta=0)
* projector    passl_v110/modeling/necks/base_neck.py:68-97 (avgpool -> fc ->
               relu -> fc), init :24-41 (kaiming normal fan_in/relu, bias 0)
* frozen BN    passl_v110/modules/freeze.py:18-23 (key encoder: global stats)

State is a flat ``dict[str, Tensor]`` using the reference's state_dict key
names and layouts: conv ``weight`` is [Cout,Cin,kh,kw];
BN keys are ``weight, bias, _mean, _variance``; ``Linear.weight`` is **[in,
out]** as in Paddle (y = x @ W + b).

[Paddle-semantics] assumptions (not checkable without Paddle, see README.md):
BatchNorm2D momentum=0.9 (running = 0.9*running + 0.1*batch), epsilon=1e-5,
running variance updated with the *biased* batch variance.
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

from .bf16 import round_act, round_weight, round_grad

BN_MOMENTUM = 0.9
BN_EPS = 1e-5
LAYERS = (3, 4, 6, 3)          # resnetimagenet.py:178 (depth 50)
PLANES = (64, 128, 256, 512)
EXPANSION = 4                  # resnetimagenet.py:113


def _bn_keys(prefix):
    return [prefix + s for s in ('.weight', '.bias', '._mean', '._variance')]


def conv_specs():
    """[(name, cout, cin, k, stride, pad)] in construction order, plus the
    name of the BN that follows each conv."""
    specs = [('conv1', 64, 3, 7, 2, 3, 'bn1')]
    inplanes = 64
    for li, (planes, blocks) in enumerate(zip(PLANES, LAYERS), start=1):
        stride = 1 if li == 1 else 2
        for b in range(blocks):
            p = 'layer%d.%d' % (li, b)
            s = stride if b == 0 else 1
            specs.append((p + '.conv1', planes, inplanes, 1, 1, 0, p + '.bn1'))
            specs.append((p + '.conv2', planes, planes, 3, s, 1, p + '.bn2'))
            specs.append((p + '.conv3', planes * 4, planes, 1, 1, 0, p + '.bn3'))
            if b == 0:  # stride != 1 or inplanes != planes*4  (resnetimagenet.py:216)
                specs.append((p + '.downsample.0', planes * 4, inplanes, 1, s, 0,
                              p + '.downsample.1'))
            inplanes = planes * 4
    return specs


def init_encoder_state(gen, in_channels=2048, hid_channels=2048, out_channels=128,
                       width_div=1, neck='NonLinearNeckV1'):
    """Returns OrderedDict with keys '0.<backbone key>' and '1.<neck key>'
    (the nn.Sequential(backbone, neck) naming of moco.py:60-61).

    ``width_div`` shrinks every channel count (tiny test configs only)."""
    st = OrderedDict()
    for name, cout, cin, k, _s, _p, bn in conv_specs():
        cout = max(cout // width_div, 1)
        cin = cin if name == 'conv1' else max(cin // width_div, 1)
        fan_out = cout * k * k
        std = math.sqrt(2.0 / fan_out)            # kaiming normal, relu gain
        st['0.' + name + '.weight'] = torch.randn(cout, cin, k, k, generator=gen) * std
        st['0.' + bn + '.weight'] = torch.ones(cout)
        st['0.' + bn + '.bias'] = torch.zeros(cout)
        st['0.' + bn + '._mean'] = torch.zeros(cout)
        st['0.' + bn + '._variance'] = torch.ones(cout)
    cin_n = in_channels // width_div
    if neck == 'LinearNeck':          # base_neck.py:44-65 (MoCo v1): avgpool + ONE fc, key `1.fc.*`
        st['1.fc.weight'] = torch.randn(cin_n, out_channels, generator=gen) * math.sqrt(2.0 / cin_n)
        st['1.fc.bias'] = torch.zeros(out_channels)
        return st
    hid = hid_channels // width_div
    # Linear [in,out]; kaiming fan_in: Paddle's fan_in for a [in,out] weight is `in`.
    st['1.mlp.0.weight'] = torch.randn(cin_n, hid, generator=gen) * math.sqrt(2.0 / cin_n)
    st['1.mlp.0.bias'] = torch.zeros(hid)
    st['1.mlp.2.weight'] = torch.randn(hid, out_channels, generator=gen) * math.sqrt(2.0 / hid)
    st['1.mlp.2.bias'] = torch.zeros(out_channels)
    return st


def trainable_keys(st):
    return [k for k in st if not (k.endswith('._mean') or k.endswith('._variance'))]


def batch_norm(x, st, prefix, use_global_stats, new_stats=None, affine_form=False):
    """paddle.nn.BatchNorm2D forward.  In training mode (use_global_stats False)
    normalises with biased batch statistics and records the running-stat update
    in ``new_stats`` (applied by the caller after the forward, functionally).
    ``affine_form`` evaluates the same function as  x*scale + shift  with
    scale = w*rsqrt(var+eps), shift = b - mean*scale  (the arithmetic order of the
    bf16 product path; used by the bf16-emulating mode only)."""
    w, b = st[prefix + '.weight'], st[prefix + '.bias']
    rm, rv = st[prefix + '._mean'], st[prefix + '._variance']
    dims = (0, 2, 3) if x.dim() == 4 else (0,)          # BatchNorm2D / BatchNorm1D on [N, C]
    if use_global_stats:
        mean, var = rm, rv
    else:
        mean = x.mean(dim=dims)
        var = x.var(dim=dims, unbiased=False)
        if new_stats is not None:
            with torch.no_grad():
                new_stats[prefix + '._mean'] = BN_MOMENTUM * rm + (1 - BN_MOMENTUM) * mean.detach()
                new_stats[prefix + '._variance'] = BN_MOMENTUM * rv + (1 - BN_MOMENTUM) * var.detach()
    inv = torch.rsqrt(var + BN_EPS)
    if affine_form:
        scale = w * inv
        shift = b - mean * scale
        if x.dim() == 2:
            return x * scale[None, :] + shift[None, :]
        return x * scale[None, :, None, None] + shift[None, :, None, None]
    if x.dim() == 2:
        return (x - mean[None, :]) * (inv * w)[None, :] + b[None, :]
    return (x - mean[None, :, None, None]) * (inv * w)[None, :, None, None] + b[None, :, None, None]


def trunk_forward(st, x, use_global_stats, new_stats=None, taps=None, maxpool=True, bf16=False,
                  frozen_stages=-1, rec=None):
    """ResNet-50 trunk (keys '0.*'): [N,3,H,W] -> layer4 map.  ``maxpool=False`` is the
    SimCLR variant (passl_v110/modeling/backbones/resnetcifar.py:275 comments the stem pool
    out, forward :321-333).

    ``bf16=True`` is the bf16-EMULATING mode (oracle/bf16.py): the same fp32 arithmetic with values
    rounded to bfloat16 exactly where the MI355X product path stores bf16 — image, operand copy of
    the weights, every conv output in training mode (BatchNorm statistics are those of the stored
    values), every BatchNorm(+residual)+ReLU output, and the same tensors' gradients on the way
    back.  With frozen-statistics BatchNorm (key encoder) the affine is applied to the unrounded
    accumulator and only the block output is rounded (conv epilogue fusion)."""
    if bf16:
        return _trunk_forward_bf16(st, x, use_global_stats, new_stats, taps, maxpool, rec)

    def conv(name, x, stride, pad):
        return F.conv2d(x, st['0.' + name + '.weight'], None, stride, pad)

    def bn(name, x):
        # resnet.py:90-106 (_freeze_stages): frozen_stages >= 0 freezes the stem's BatchNorm, and
        # layer1..layer<frozen_stages> — their BatchNorms use the running statistics
        stage = 0 if not name.startswith('layer') else int(name[5])
        frozen = frozen_stages >= 0 and stage <= frozen_stages
        return batch_norm(x, st, '0.' + name, use_global_stats or frozen, new_stats)

    x = F.relu(bn('bn1', conv('conv1', x, 2, 3)))
    if maxpool:
        x = F.max_pool2d(x, 3, 2, 1)
    if taps is not None:
        taps['stem'] = x
    for li, blocks in enumerate(LAYERS, start=1):
        for b in range(blocks):
            p = 'layer%d.%d' % (li, b)
            s = 2 if (li > 1 and b == 0) else 1
            identity = x
            out = F.relu(bn(p + '.bn1', conv(p + '.conv1', x, 1, 0)))
            out = F.relu(bn(p + '.bn2', conv(p + '.conv2', out, s, 1)))
            out = bn(p + '.bn3', conv(p + '.conv3', out, 1, 0))
            if b == 0:
                identity = bn(p + '.downsample.1', conv(p + '.downsample.0', x, s, 0))
            x = F.relu(out + identity)
        if taps is not None:
            taps['layer%d' % li] = x
    return x


# Second valid evaluation of the bf16 contract: products of bf16 operands accumulated in float64
# instead of float32 (summation order / accumulator width are NOT part of the contract).  The golden
# generator runs both; their difference is how far two correct bf16 implementations may be apart at a
# given (ill-conditioned, random-init) point, and the GPU tests scale their bounds with it.
ACCUM64 = False


def _conv(x, w, stride, pad):
    if ACCUM64:
        return F.conv2d(x.double(), w.double(), None, stride, pad).float()
    return F.conv2d(x, w, None, stride, pad)


def _matmul(x, w):
    if ACCUM64:
        return (x.double() @ w.double()).float()
    return x @ w


def _trunk_forward_bf16(st, x, use_global_stats, new_stats, taps, maxpool, rec=None):
    """trunk_forward with the product path's bf16 storage points (see trunk_forward).

    Training mode (query encoder): conv output rounded (stored, BatchNorm statistics are those of the
    stored values); BatchNorm + residual + ReLU evaluated in fp32 and rounded once; the gradient
    entering a block through conv1 / the downsample conv is rounded before the residual-fork sum
    (the data-gradient kernel stores it in bf16 tiles before adding the other branch's gradient).
    Frozen statistics (key encoder): one kernel per conv — affine on the accumulator, rounded, THEN
    the residual add + ReLU, rounded again (conv epilogue order)."""
    train = not use_global_stats

    def keep(key, t):
        """``rec`` (per-layer teacher forcing, tests/test_layers_gpu.py): every stored tensor of the training-mode
        forward is kept with its gradient retained — after backward, ``t.grad`` is the gradient w.r.t. the STORED
        tensor in fp32 (what the consuming backward kernel rounds to bf16 and reads)."""
        if rec is not None and t.requires_grad:
            t.retain_grad()
        if rec is not None:
            rec[key] = t
        return t

    def conv(name, x, stride, pad):
        keep(name + '.x', x)
        y = _conv(x, round_weight(st['0.' + name + '.weight']), stride, pad)
        return keep(name + '.y', round_act(y)) if train else y       # training: the conv output is stored (bf16)

    def bn(name, x):
        return batch_norm(x, st, '0.' + name, use_global_stats, new_stats, affine_form=True)

    x = round_act(x)                              # bf16 NHWC image
    x = keep('bn1.z', round_act(F.relu(bn('bn1', conv('conv1', x, 2, 3)))))
    if maxpool:
        x = F.max_pool2d(x, 3, 2, 1)
        if train:
            x = round_grad(x)                      # the pooled map's gradient is a stored bf16 tensor
        keep('maxpool.z', x)
    if taps is not None:
        taps['stem'] = x
    for li, blocks in enumerate(LAYERS, start=1):
        for b in range(blocks):
            p = 'layer%d.%d' % (li, b)
            s = 2 if (li > 1 and b == 0) else 1
            identity = x
            # every data-gradient launch stores bf16 tiles BEFORE the fork gradients meet: conv1's and
            # the downsample conv's input gradients are rounded separately, then summed, then rounded
            xin = round_grad(x) if train else x
            xds = round_grad(x) if train else x
            out = keep(p + '.bn1.z', round_act(F.relu(bn(p + '.bn1', conv(p + '.conv1', xin, 1, 0)))))
            out = keep(p + '.bn2.z', round_act(F.relu(bn(p + '.bn2', conv(p + '.conv2', out, s, 1)))))
            out = bn(p + '.bn3', conv(p + '.conv3', out, 1, 0))
            if not train:
                out = round_act(out)               # epilogue: affine -> bf16 tile -> + residual -> ReLU -> bf16
            if b == 0:
                identity = keep(p + '.downsample.1.z',
                                round_act(bn(p + '.downsample.1', conv(p + '.downsample.0', xds, s, 0))))
            elif rec is not None:
                identity = x.clone()               # own autograd node: its gradient is the residual branch's alone
            keep(p + '.bn3.res', identity)
            x = keep(p + '.bn3.z', round_act(F.relu(out + identity)))
        if taps is not None:
            taps['layer%d' % li] = x
    return x


def encoder_forward(st, x, use_global_stats, new_stats=None, taps=None, bf16=False):
    """nn.Sequential(ResNet(depth=50,num_classes=0,with_pool=False),
    NonLinearNeckV1(...)) forward; ``st`` keys as in init_encoder_state.
    ``bf16``: bf16-emulating mode (trunk_forward): pooled features and the hidden layer are stored
    in bf16, the projector output is fp32 but its gradient enters the Linear backward rounded."""
    x = trunk_forward(st, x, use_global_stats, new_stats, taps, maxpool=True, bf16=bf16)
    # NonLinearNeckV1.forward (base_neck.py:93-97)
    x = F.adaptive_avg_pool2d(x, 1).reshape(x.shape[0], -1)
    if bf16:
        x = round_act(x)
        if '1.fc.weight' in st:
            return round_grad(_matmul(x, round_weight(st['1.fc.weight'])) + st['1.fc.bias'])
        x = round_act(F.relu(_matmul(x, round_weight(st['1.mlp.0.weight'])) + st['1.mlp.0.bias']))
        return round_grad(_matmul(x, round_weight(st['1.mlp.2.weight'])) + st['1.mlp.2.bias'])
    if '1.fc.weight' in st:           # LinearNeck.forward (base_neck.py:61-65)
        return x @ st['1.fc.weight'] + st['1.fc.bias']
    x = F.relu(x @ st['1.mlp.0.weight'] + st['1.mlp.0.bias'])
    x = x @ st['1.mlp.2.weight'] + st['1.mlp.2.bias']
    return x
