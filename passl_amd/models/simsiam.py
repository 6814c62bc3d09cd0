"""SimSiam (ResNet-50) pre-training on the MI355X HIP path — reference passl/models/simsiam.py.

Constructor arguments, factory name, sub-layer / state_dict names and the ``model([x1, x2]) -> loss`` contract are the
reference's: ``SimSiamPretain`` :36-95 (``encoder`` = v2 ResNet whose ``fc`` becomes the 3-layer projector
[Linear(no bias) - BatchNorm1D - ReLU] x 2 - the original fc - BatchNorm1D(no gamma / beta), the original fc's bias
kept but without gradient; ``predictor`` = Linear(no bias) - BatchNorm1D - ReLU - Linear), factory :152-163.

Execution: the trunk is the MoCo hot path's (implicit-GEMM convs with fused BatchNorm statistics, fused BatchNorm
backward in the data-gradient epilogues, weight gradients on the side stream); each view is its own pass (BatchNorm
statistics are per view, as in the reference); MLPs = GEMM kernel with fp32 output -> BatchNorm1D on fp32 rows -> ReLU
(SimCLR-neck pattern); the criterion is one fused kernel pair (passl.loss.simsiam).  Parameter groups of the task
yaml (``encoder`` on the schedule, ``predictor`` at a fixed rate) map to two EncoderArenas = two flat optimizer
launches.  The reference converts every BatchNorm to SyncBatchNorm when world_size > 1 (:160-162): here the
BatchNorm layers then fold their slabs to fp64 moments, all-gather them (3 C doubles per layer and rank) and combine
them in rank order (csrc/bn.hip: bn_moments / bn_finalize_moments, bn_bwd_sums / bn_bwd_finalize_sums) — a two-rank run
equals the one-rank run on the concatenated batch (tests/test_dp_gpu.py).
"""
import os
import pickle
from functools import partial

import torch
import torch.nn as tnn

from ..core.sync_utils import collectives_active
from ..hip import config
from ..hip import nn as hnn
from ..hip.nn import EncoderArena
from ..loss.simsiam import neg_cosine_similarity
from ..utils.checkpoint import load_lenient, load_pickle, to_numpy
from .base_model import Model
from ..modeling.backbones.resnet import ResNet as _Trunk
from .resnet import BottleneckBlock, ResNet, paddle_default_linear_init_

__all__ = ['SimSiamPretain', 'SimSiamLinearProbe', 'simsiam_resnet50_pretrain', 'simsiam_resnet50_linearprobe']


class _MLP(tnn.Sequential):
    """Linear -> (BatchNorm1D -> (ReLU)) chains with the reference's Sequential indices as sub-layer names; Linears
    write fp32, BatchNorm works on fp32 rows, the next Linear reads the compute-dtype cast."""

    def forward(self, x):
        dt = config.get_compute_dtype()
        mods = list(self)
        i = 0
        while i < len(mods):
            lin = mods[i]
            x = lin(hnn.to_compute(x, dt), out_f32=True)
            i += 1
            if i < len(mods) and isinstance(mods[i], hnn._BatchNormBase):
                relu = i + 1 < len(mods) and isinstance(mods[i + 1], hnn.ReLU)
                x = mods[i](x, relu=relu)
                i += 2 if relu else 1
        return x


class SimSiamPretain(Model):
    """Build a SimSiam Pretrain model."""

    def __init__(self, base_encoder, dim=2048, pred_dim=512):
        super().__init__()
        # create the encoder; num_classes is the output fc dimension, zero-initialize last BNs
        self.encoder = base_encoder(class_num=dim, zero_init_residual=True)
        # build a 3-layer projector
        fc = self.encoder.fc
        prev_dim = fc.weight.shape[1]
        self.encoder.fc = _MLP(hnn.Linear(prev_dim, prev_dim, bias_attr=False), hnn.BatchNorm1D(prev_dim), hnn.ReLU(),
                               hnn.Linear(prev_dim, prev_dim, bias_attr=False), hnn.BatchNorm1D(prev_dim), hnn.ReLU(),
                               fc, hnn.BatchNorm1D(dim, weight_attr=False, bias_attr=False))
        self.encoder.fc[6].bias.requires_grad_(False)        # hack: not use bias as it is followed by BN
        # build a 2-layer predictor
        self.predictor = _MLP(hnn.Linear(dim, pred_dim, bias_attr=False), hnn.BatchNorm1D(pred_dim), hnn.ReLU(),
                              hnn.Linear(pred_dim, dim))
        for mlp in (self.encoder.fc, self.predictor):       # nn.Linear's framework default (the original fc has it already)
            for layer in mlp:
                if isinstance(layer, hnn.Linear) and layer is not fc:
                    paddle_default_linear_init_(layer)
        self.arena_q = EncoderArena(self.encoder, trainable=True, exclude_params=[self.encoder.fc[6].bias])
        self.arena_p = EncoderArena(self.predictor, trainable=True)

    # -- state plumbing
    def trainable_arenas(self):
        return [self.arena_q, self.arena_p]

    def sync_runtime_state(self):
        self.arena_q.refresh()
        self.arena_p.refresh()

    def load_state_dict(self, state_dict, strict=True):
        r = super().load_state_dict(state_dict, strict=strict)
        self.sync_runtime_state()
        return r

    # -- reference API
    def _view(self, x):
        y = _Trunk.forward(self.encoder, x)                  # trunk + average pool (NHWC [N, 1, 1, 2048])
        z = self.encoder.fc(y.reshape(y.shape[0], -1))
        return z, self.predictor(z)

    def forward(self, inputs):
        assert isinstance(inputs, (list, tuple))
        x1, x2 = inputs[0], inputs[1]
        self.arena_q.refresh()                   # compute-dtype operands from the fp32 masters (after the update)
        self.arena_p.refresh()
        z1, p1 = self._view(x1)                  # compute features for one view: NxC
        z2, p2 = self._view(x2)
        return (neg_cosine_similarity(p1, z2) + neg_cosine_similarity(p2, z1)) * 0.5

    def load_pretrained(self, path, rank=0, finetune=False):
        if not os.path.exists(path + '.pdparams'):
            raise ValueError('Model pretrain path {} does not exists.'.format(path))
        load_lenient(self, load_pickle(path + '.pdparams'), what='pretrained model')
        self.sync_runtime_state()

    def save(self, path, local_rank=0, rank=0):
        """<path>.pdparams = the whole state; <path>_encoder.pdparams = the trunk without the projector, prefix
        removed (simsiam.py:113-126)."""
        if rank != 0:
            return
        os.makedirs(os.path.dirname(path) or '.', exist_ok=True)
        sd = to_numpy(dict(self.state_dict()))
        with open(path + '.pdparams', 'wb') as f:
            pickle.dump(sd, f, protocol=2)
        enc = {k[len('encoder.'):]: v for k, v in sd.items() if k.startswith('encoder') and not k.startswith('encoder.fc')}
        with open(path + '_encoder.pdparams', 'wb') as f:
            pickle.dump(enc, f, protocol=2)


class SimSiamLinearProbe(ResNet):
    """simsiam.py:128-147: a v2 ResNet whose every parameter but ``fc.weight`` / ``fc.bias`` is frozen and whose every
    BatchNorm uses its running statistics also in train mode (``_use_global_stats``); the classifier starts at
    Normal(0, 0.01) / zero bias.  ``load_pretrained`` takes the ``<prefix>_encoder.pdparams`` file ``SimSiamPretain.save``
    writes (trunk keys without the ``encoder.`` prefix; the absent ``fc`` keeps its initialisation).

    Execution: the frozen trunk lives in a NON-trainable arena and runs the fused inference path (BatchNorm + ReLU +
    residual folded into the conv epilogues: one kernel per conv, nothing kept for backward); the classifier is the
    only trainable arena — one GEMM with fp32 scores forward, its weight / bias gradients backward."""

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        # freeze all layers but the last fc
        self.frozen_stages = 4
        self._freeze_stages()
        # optimize only the linear classifier
        parameters = [p for p in self.parameters() if p.requires_grad]
        assert len(parameters) == 2  # weight, bias
        with torch.no_grad():
            self.fc.weight.copy_(torch.randn(self.fc.weight.shape) * 0.01)
            self.fc.bias.zero_()
        self.arena_k = EncoderArena(tnn.ModuleList(self.frozen_modules()), trainable=False)   # (name as in MoCo)
        self.arena_k.update_bn_affine()
        self.arena_q = EncoderArena(self.fc, trainable=True)       # what the optimizer / reducer see

    def sync_runtime_state(self):
        """After weights were written from outside (checkpoint / pre-trained trunk)."""
        self.arena_k.refresh()
        self.arena_k.update_bn_affine()
        self.arena_q.refresh()

    def load_state_dict(self, state_dict, strict=True):
        r = super().load_state_dict(state_dict, strict=strict)
        self.sync_runtime_state()
        return r

    def load_pretrained(self, path, rank=0, finetune=False):
        super().load_pretrained(path, rank=rank, finetune=finetune)
        self.sync_runtime_state()

    def forward(self, x):
        self.arena_q.refresh()                  # compute-dtype classifier operands from the fp32 masters
        return super().forward(x)


def simsiam_resnet50_pretrain(**kwargs):
    encoder = partial(ResNet, block=BottleneckBlock, depth=50)
    model = SimSiamPretain(base_encoder=encoder, dim=2048, pred_dim=512, **kwargs)
    # Apply SyncBN (simsiam.py:160-162)
    if collectives_active():
        hnn.convert_sync_batchnorm(model)
    return model


def simsiam_resnet50_linearprobe(**kwargs):
    return SimSiamLinearProbe(block=BottleneckBlock, depth=50, **kwargs)
