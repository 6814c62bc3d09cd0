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


def batch_norm(x, st, prefix, use_global_stats, new_stats=None):
    """paddle.nn.BatchNorm2D forward.  In training mode (use_global_stats False)
    normalises with biased batch statistics and records the running-stat update
    in ``new_stats`` (applied by the caller after the forward, functionally)."""
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
    if x.dim() == 2:
        return (x - mean[None, :]) * (inv * w)[None, :] + b[None, :]
    return (x - mean[None, :, None, None]) * (inv * w)[None, :, None, None] + b[None, :, None, None]


def trunk_forward(st, x, use_global_stats, new_stats=None, taps=None, maxpool=True):
    """ResNet-50 trunk (keys '0.*'): [N,3,H,W] -> layer4 map.  ``maxpool=False`` is the
    SimCLR variant (passl_v110/modeling/backbones/resnetcifar.py:275 comments the stem pool
    out, forward :321-333)."""
    def conv(name, x, stride, pad):
        return F.conv2d(x, st['0.' + name + '.weight'], None, stride, pad)

    def bn(name, x):
        return batch_norm(x, st, '0.' + name, use_global_stats, new_stats)

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


def encoder_forward(st, x, use_global_stats, new_stats=None, taps=None):
    """nn.Sequential(ResNet(depth=50,num_classes=0,with_pool=False),
    NonLinearNeckV1(...)) forward; ``st`` keys as in init_encoder_state."""
    x = trunk_forward(st, x, use_global_stats, new_stats, taps, maxpool=True)
    # NonLinearNeckV1.forward (base_neck.py:93-97)
    x = F.adaptive_avg_pool2d(x, 1).reshape(x.shape[0], -1)
    if '1.fc.weight' in st:           # LinearNeck.forward (base_neck.py:61-65)
        return x @ st['1.fc.weight'] + st['1.fc.bias']
    x = F.relu(x @ st['1.mlp.0.weight'] + st['1.mlp.0.bias'])
    x = x @ st['1.mlp.2.weight'] + st['1.mlp.2.bias']
    return x
