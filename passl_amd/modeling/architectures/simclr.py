"""SimCLR on the MI355X HIP path — reference passl_v110/modeling/architectures/simclr.py:30-76.

train_iter = concat the two views -> ONE encoder pass over 2N images (train-mode BN statistics
over both views, as in the reference) -> l2_normalize -> split -> SimCLRContrastiveHead.
Attribute names (``encoder``, ``backbone``, ``head``) and the ``forward(*inputs, mode=...)``
contract are the reference's; the encoder's parameters live in one EncoderArena (flat fp32 buffer
+ flat gradient buffer) so LARS is a two-launch multi-tensor update."""
import torch

from ...hip import nn, ops
from ...hip.nn import EncoderArena
from ..backbones import build_backbone
from ..heads import build_head
from ..necks import build_neck
from .builder import MODELS


class _SplitViews(torch.autograd.Function):
    """(con[:n], con[n:]) — the two views' rows.  Backward writes the two gradients into ONE buffer with the library's
    copy kernel; autograd's own slice backward would be two zero fills, two copies and an add per step."""

    @staticmethod
    def forward(ctx, con, n):
        ctx.n, ctx.rows = n, con.shape[0]
        ctx.set_materialize_grads(False)
        return con[:n], con[n:]

    @staticmethod
    def backward(ctx, dq, dk):
        ref = dq if dq is not None else dk
        if ref is None:
            return None, None
        n = ctx.n
        out = torch.empty((ctx.rows,) + tuple(ref.shape[1:]), dtype=ref.dtype, device=ref.device)
        for part, g in ((out[:n], dq), (out[n:], dk)):
            if g is None:
                ops.fill_zero(part) if part.is_cuda else part.zero_()
            elif part.is_cuda:
                ops.copy_into(part, g.contiguous())
            else:
                part.copy_(g)
        return out, None


@MODELS.register()
class SimCLR(nn.Layer):
    @property
    def graph_safe(self):
        """Replayable from a recorded native plan (hip/replay.py) unless the head gathers embeddings across ranks
        (those collectives sit inside the loss function's forward / backward)."""
        return not getattr(self.head, 'multi_rank', False)

    def __init__(self, backbone, neck=None, head=None, dim=128, T=0.5):
        super().__init__()
        self.T = T
        self.encoder = torch.nn.Sequential(build_backbone(backbone), build_neck(neck))
        self.backbone = self.encoder[0]
        self.head = build_head(head)
        # ``frozen_stages >= 0`` (configs/simclr/simclr_r18_cifar10.yaml:8 freezes the whole trunk): the frozen prefix
        # lives in a non-trainable arena and runs the fused inference path, the rest shares the trainable arena with
        # the projector — the split of architectures/clas.py
        bb = self.backbone
        frozen = bb.frozen_modules() if hasattr(bb, 'frozen_modules') else []
        self.arena_k = None
        if frozen:
            self.arena_k = EncoderArena(torch.nn.ModuleList(frozen), trainable=False)
            self.arena_k.update_bn_affine()
            object.__setattr__(self, '_live', torch.nn.ModuleList(list(bb.trainable_modules()) + [self.encoder[1]]))
            self.arena_q = EncoderArena(self._live, trainable=True)
        else:
            self.arena_q = EncoderArena(self.encoder, trainable=True)    # name shared with MoCo (DP reducer)

    def load_state_dict(self, state_dict, strict=True):
        r = super().load_state_dict(state_dict, strict=strict)
        self.sync_runtime_state()
        return r

    def sync_runtime_state(self):
        if self.arena_k is not None:
            self.arena_k.refresh()
            self.arena_k.update_bn_affine()
        self.arena_q.refresh()

    def train_iter(self, *inputs, **kwargs):
        img_q, img_k = inputs
        self.arena_q.refresh()
        n = img_q.shape[0]
        if img_q.is_cuda and img_q.is_contiguous() and img_k.is_contiguous() and img_q.dtype == img_k.dtype:
            # paddle.concat([img_q, img_k]) with the library's copy kernel (no framework launch inside the step)
            img_con = torch.empty((2 * n,) + tuple(img_q.shape[1:]), dtype=img_q.dtype, device=img_q.device)
            ops.copy_into(img_con[:n], img_q)
            ops.copy_into(img_con[n:], img_k)
        else:
            img_con = torch.cat([img_q, img_k])
        con = self.encoder(img_con)
        con = nn.normalize(con, axis=1)                    # layers.l2_normalize(con, -1)
        q, k = _SplitViews.apply(con, n)
        return self.head(q, k)

    def forward(self, *inputs, mode='train', **kwargs):
        if mode == 'train':
            return self.train_iter(*inputs, **kwargs)
        elif mode == 'extract':
            with torch.no_grad():
                self.arena_q.refresh()
                return self.backbone(*inputs)
        elif mode == 'test':
            raise NotImplementedError('SimCLR.test_iter is broken in the reference (simclr.py:62-68 '
                                      'calls an undefined backbone_forward)')
        else:
            raise Exception('No such mode: {}'.format(mode))
