"""Classification (linear probe / supervised head on a backbone) — reference
passl_v110/modeling/architectures/clas.py:25-77: ``train_iter(img, label)`` = backbone -> head ->
``head.loss``; ``test_iter`` returns the class scores; modes train / test / extract / infer.

HIP execution: the frozen part of the trunk (``backbone.frozen_stages``, resnet.py:90-106; 4 = all of it
for the linear-probe configs, configs/moco/moco_clas_r50.yaml) lives in a non-trainable EncoderArena and
runs the fused inference path (BatchNorm with running statistics + ReLU + residual folded into the conv
epilogues: one kernel per conv); the trainable remainder (layer<frozen_stages+1>..layer4, or the whole
trunk for frozen_stages -1) shares ONE trainable arena with the head — what the optimizer and the
data-parallel reducer see."""
import torch

from ...hip import nn
from ...hip.nn import EncoderArena
from ..backbones import build_backbone
from ..heads import build_head
from .builder import MODELS


@MODELS.register()
class Classification(nn.Layer):
    """Simple image classification."""

    def __init__(self, backbone, with_sobel=False, head=None):
        super(Classification, self).__init__()
        if with_sobel:
            raise NotImplementedError('with_sobel is a TODO in the reference (clas.py:35-37)')
        self.with_sobel = with_sobel
        self.backbone = build_backbone(backbone)
        if head is None:
            raise NotImplementedError('Classification without a head')
        self.head = build_head(head)
        frozen = self.backbone.frozen_modules() if hasattr(self.backbone, 'frozen_modules') else []
        live = self.backbone.trainable_modules() if hasattr(self.backbone, 'trainable_modules') else []
        # arenas over plain lists of sub-layers (containers that do not re-parent the modules)
        self.arena_k = None
        if frozen:
            self.arena_k = EncoderArena(torch.nn.ModuleList(frozen), trainable=False)   # name as in MoCo
            self.arena_k.update_bn_affine()
        object.__setattr__(self, '_live', torch.nn.ModuleList(list(live) + [self.head]))
        self.arena_q = EncoderArena(self._live, trainable=True)        # what the optimizer / reducer see

    def load_state_dict(self, state_dict, strict=True):
        r = super().load_state_dict(state_dict, strict=strict)
        self.sync_runtime_state()
        return r

    def sync_runtime_state(self):
        """Call after weights were written from outside (checkpoint / pretrained backbone load)."""
        if self.arena_k is not None:
            self.arena_k.refresh()
            self.arena_k.update_bn_affine()
        self.arena_q.refresh()

    def backbone_forward(self, x):
        return self.backbone(x)

    def train_iter(self, *inputs, **kwargs):
        img, label = inputs
        self.arena_q.refresh()
        x = self.backbone_forward(img)
        outs = self.head(x)
        return self.head.loss(outs, label)

    def test_iter(self, *inputs, **kwargs):
        with torch.no_grad():
            img, label = inputs
            self.arena_q.refresh()
            return self.head(self.backbone_forward(img))

    def infer_iter(self, *inputs, **kwargs):
        with torch.no_grad():
            self.arena_q.refresh()
            return self.head(self.backbone_forward(*inputs))

    def forward(self, *inputs, mode='train', **kwargs):
        if mode == 'train':
            return self.train_iter(*inputs, **kwargs)
        elif mode == 'test':
            return self.test_iter(*inputs, **kwargs)
        elif mode == 'extract':
            return self.backbone(*inputs)
        elif mode == 'infer':
            return self.infer_iter(*inputs)
        else:
            raise Exception("No such mode: {}".format(mode))
