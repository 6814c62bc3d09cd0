"""Weight initialisers with the reference's fan conventions (passl_v110/modules/init.py:24-55,
250-330): 2-D weights are [in, out] (fan_in = shape[0]); conv weights [Cout,Cin,kh,kw]."""
import math

import torch


def _calculate_fan_in_and_fan_out(tensor):
    if tensor.dim() < 2:
        raise ValueError('Fan in and fan out can not be computed for tensor with fewer than 2 dimensions')
    if tensor.dim() == 2:
        num_in, num_out = tensor.shape[0], tensor.shape[1]
    else:
        num_in, num_out = tensor.shape[1], tensor.shape[0]
    rf = 1
    if tensor.dim() > 2:
        rf = tensor[0][0].numel()
    return num_in * rf, num_out * rf


def calculate_gain(nonlinearity, param=None):
    if nonlinearity in ('linear', 'conv1d', 'conv2d', 'conv3d', 'sigmoid'):
        return 1
    if nonlinearity == 'tanh':
        return 5.0 / 3
    if nonlinearity == 'relu':
        return math.sqrt(2.0)
    if nonlinearity == 'leaky_relu':
        slope = 0.01 if param is None else param
        return math.sqrt(2.0 / (1 + slope ** 2))
    raise ValueError('Unsupported nonlinearity {}'.format(nonlinearity))


@torch.no_grad()
def constant_(x, value):
    x.fill_(value)
    return x


@torch.no_grad()
def normal_(x, mean=0., std=1.):
    x.copy_(torch.randn(x.shape) * std + mean)
    return x


@torch.no_grad()
def kaiming_normal_(x, a=0, mode='fan_in', nonlinearity='leaky_relu'):
    fan_in, fan_out = _calculate_fan_in_and_fan_out(x)
    fan = fan_in if mode == 'fan_in' else fan_out
    std = calculate_gain(nonlinearity, a) / math.sqrt(fan)
    # drawn on the host with the global torch RNG (seeded by Trainer like paddle.seed)
    x.copy_(torch.randn(x.shape) * std)
    return x


def kaiming_init(layer, a=0, mode='fan_out', nonlinearity='relu', bias=0, distribution='normal'):
    assert distribution == 'normal'
    kaiming_normal_(layer.weight, a=a, mode=mode, nonlinearity=nonlinearity)
    if getattr(layer, 'bias', None) is not None:
        constant_(layer.bias, bias)


def constant_init(layer, val, bias=0):
    if getattr(layer, 'weight', None) is not None:
        constant_(layer.weight, val)
    if getattr(layer, 'bias', None) is not None:
        constant_(layer.bias, bias)


def normal_init(layer, mean=0, std=1, bias=0):
    normal_(layer.weight, mean, std)
    if getattr(layer, 'bias', None) is not None:
        constant_(layer.bias, bias)
