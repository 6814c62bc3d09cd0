"""MAE_PRETRAIN wrapper — reference passl_v110/modeling/architectures/MAE.py:30-55.

The reference's ``train_iter`` hands the whole ``(img, label)`` tuple to the backbone and returns
the ``(loss, pred, mask)`` tuple, which OptimizerHook cannot index (``outputs['loss']``,
hooks/optimizer_hook.py:32) — the wrapper as shipped never trains.  Here it takes
``inputs[0]`` and returns ``{'loss': loss}`` so that configs/mae/mae_vit_b_pretrain.yaml runs
unchanged through the v110 Trainer."""
import torch

from ...hip import nn
from ...hip.nn import EncoderArena
from ..backbones import build_backbone
from ..heads import build_head
from .builder import MODELS


@MODELS.register()
class MAE_PRETRAIN(nn.Layer):
    # replayable from a recorded native plan (hip/replay.py): the masking noise is drawn by a live host call between
    # two plan segments (backbones/mae.py:random_masking_ids), everything else that varies sits in device memory
    graph_safe = True

    def __init__(self, architecture=None, mask_ratio=0.75):
        super().__init__()
        self.backbone = build_backbone(architecture)
        self.mask_ratio = mask_ratio
        self.arena_q = EncoderArena(self.backbone, trainable=True)    # flat params/grads (AdamW, DP)

    def load_state_dict(self, state_dict, strict=True):
        r = super().load_state_dict(state_dict, strict=strict)
        self.arena_q.refresh()
        return r

    def train_iter(self, *inputs, **kwargs):
        img = inputs[0]
        self.arena_q.refresh()
        loss, pred, mask = self.backbone(img, self.mask_ratio, noise=kwargs.get('noise', None))
        return dict(loss=loss)

    def forward(self, *inputs, mode='train', **kwargs):
        if mode == 'train':
            return self.train_iter(*inputs, **kwargs)
        elif mode == 'extract':
            with torch.no_grad():
                self.arena_q.refresh()
                return self.backbone(*inputs)
        else:
            raise Exception('No such mode: {}'.format(mode))


@MODELS.register()
class MAE_FINETUNE(nn.Layer):
    """Fine-tuning wrapper — reference passl_v110/modeling/architectures/MAE.py:58-94: ``train_iter(img, label)`` =
    backbone -> head -> ``head.loss`` (loss / acc1 / acc5); ``extract`` returns the pooled features.  The reference
    routes ``mode='test'`` to a ``test_iter`` it never defines (MAE.py:88-89); here it returns the class scores like
    Classification.test_iter.  Backbone and head share one trainable arena (AdamW over flat buffers, DP reducer)."""

    def __init__(self, architecture=None, head=None):
        super().__init__()
        self.backbone = build_backbone(architecture)
        self.head = build_head(head)
        object.__setattr__(self, '_live', torch.nn.ModuleList([self.backbone, self.head]))
        self.arena_q = EncoderArena(self._live, trainable=True)

    def load_state_dict(self, state_dict, strict=True):
        r = super().load_state_dict(state_dict, strict=strict)
        self.arena_q.refresh()
        return r

    def sync_runtime_state(self):
        self.arena_q.refresh()

    def backbone_forward(self, x):
        return self.backbone(x)

    def train_iter(self, *inputs, **kwargs):
        img, label = inputs
        self.arena_q.refresh()
        outs = self.head(self.backbone_forward(img))
        return self.head.loss(outs, label)

    def test_iter(self, *inputs, **kwargs):
        with torch.no_grad():
            self.arena_q.refresh()
            return self.head(self.backbone_forward(inputs[0]))

    def forward(self, *inputs, mode='train', **kwargs):
        if mode == 'train':
            return self.train_iter(*inputs, **kwargs)
        elif mode == 'test':
            return self.test_iter(*inputs, **kwargs)
        elif mode == 'extract':
            with torch.no_grad():
                self.arena_q.refresh()
                return self.backbone(*inputs)
        else:
            raise Exception('No such mode: {}'.format(mode))
