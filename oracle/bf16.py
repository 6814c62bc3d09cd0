"""bf16-emulating building blocks of the oracle (TEST INFRASTRUCTURE, see oracle/__init__.py).

The reference has no bf16 MoCo path (its MoCo is fp32, optionally fp16 AMP), so
there is nothing in /root/reference to restate here.  What this file pins instead is the PRODUCT's
numerical contract in bf16 mode: "the reference's fp32 algorithm (oracle/resnet50.py, oracle/moco.py,
both pinned by executing the reference's own sources) with values rounded to bfloat16
(round-to-nearest-even) at the tensors the MI355X path stores in bf16, fp32 accumulation everywhere".
A product of two bfloat16 values is exact in fp32, so an fp32 CPU convolution over bf16-rounded
operands IS a bf16-operand / fp32-accumulate convolution up to summation order.

Three straight-through nodes:
  round_act(x)     value AND gradient are rounded   — an activation tensor stored in bf16, whose
                   gradient tensor is stored in bf16 as well
  round_weight(w)  value rounded, gradient passed   — the bf16 operand copy of an fp32 master weight
                   (the weight gradient is accumulated and kept in fp32)
  round_grad(x)    value passed, gradient rounded   — an fp32 output whose gradient the next
                   backward kernel consumes in bf16 (the projector output)
"""
import torch
from torch.autograd import Function


def _rne(t):
    return t.to(torch.bfloat16).to(t.dtype)


class _RoundAct(Function):
    @staticmethod
    def forward(ctx, x):
        return _rne(x)

    @staticmethod
    def backward(ctx, g):
        return _rne(g)


class _RoundWeight(Function):
    @staticmethod
    def forward(ctx, w):
        return _rne(w)

    @staticmethod
    def backward(ctx, g):
        return g


class _RoundGrad(Function):
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return _rne(g)


def round_act(x):
    return _RoundAct.apply(x)


def round_weight(w):
    return _RoundWeight.apply(w)


def round_grad(x):
    return _RoundGrad.apply(x)
