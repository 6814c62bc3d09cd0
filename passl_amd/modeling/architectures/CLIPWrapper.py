"""CLIPWrapper — reference passl_v110/modeling/architectures/CLIPWrapper.py:26-60: ``train_iter(image,
text)`` builds labels ``arange(B)``, calls ``self.model(image, text, is_train=True)`` and hands the
two logits matrices to the head (CLIPHead).  Parameters live in one EncoderArena (flat fp32 master /
gradient buffers for AdamW and the data-parallel reducer)."""
import torch
import torch.distributed as dist

from ...core.sync_utils import collectives_active
from ...hip import nn
from ...hip.nn import EncoderArena
from ..backbones import build_backbone
from ..heads import build_head
from .builder import MODELS


@MODELS.register()
class CLIPWrapper(nn.Layer):
    def __init__(self, architecture=None, head=None, multi_rank=False):
        """multi_rank (extension, default off = the reference): the contrastive loss sees the features of
        every data-parallel rank (BASELINE configs[4] "cross-GPU InfoNCE"): all-gather over RCCL, labels
        ``arange(B) + B*rank`` as in passl/models/mocov3.py:187-198."""
        super().__init__()
        self.multi_rank = bool(multi_rank)
        self.model = build_backbone(architecture)
        self.automatic_optimization = False
        self.head = build_head(head)
        self.arena_q = EncoderArena(self.model, trainable=True)

    def load_state_dict(self, state_dict, strict=True):
        r = super().load_state_dict(state_dict, strict=strict)
        self.arena_q.refresh()
        return r

    @property
    def graph_safe(self):
        """May the step be replayed from a recorded native plan (hip/replay.py)?  Yes: nothing varies from step to step
        on the host; the cross-rank extension's collectives stay live host calls between plan segments."""
        return True

    def _arange(self, n, device, off=0):
        """arange(n) + off on `device`, built once (a fresh one per step would be an ATen kernel inside the step)."""
        cache = self.__dict__.setdefault('_label_cache', {})
        lab = cache.get((n, str(device), off))
        if lab is None:
            lab = cache[(n, str(device), off)] = torch.arange(n, device=device) + off
            lab._passl_is_arange = off == 0
        return lab

    def train_iter(self, *inputs, **kwargs):
        image, text = inputs
        # the reference's labels are arange(len(image)); CLIPHead's kernel has them built in
        img_labels = self._arange(len(image), image.device)
        text_labels = self._arange(len(text), text.device)
        self.arena_q.refresh()
        if self.multi_rank:
            off = len(image) * dist.get_rank() if collectives_active() else 0
            img_logits, text_logits = self.model(image, text, is_train=True, multi_rank=True)
            return self.head(img_logits, text_logits, self._arange(len(image), image.device, off),
                             self._arange(len(text), text.device, off))
        img_logits, text_logits = self.model(image, text, is_train=True)
        return self.head(img_logits, text_logits, img_labels, text_labels)

    def forward(self, *inputs, mode='train', **kwargs):
        if mode == 'train':
            return self.train_iter(*inputs, **kwargs)
        elif mode == 'extract':
            with torch.no_grad():
                self.arena_q.refresh()
                return self.model.encode_image(inputs[0])
        else:
            raise Exception("No such mode: {}".format(mode))
