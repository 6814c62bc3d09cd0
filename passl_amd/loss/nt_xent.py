"""``passl.loss.nt_xent`` — SimCLR's NT-Xent (+ 3 x CO2 consistency) loss, reference
passl_v110/modeling/heads/simclr_contrastive_head.py:42-102.

    loss, acc1 = nt_xent(h1, h2, T, co2_weight=3.0, gather=False)

h1, h2: the two views' L2-normalised embeddings [N,128] fp32.  ONE fused HIP kernel each way
(csrc/ntxent.hip): the N x 2N logit matrices never reach HBM.  ``gather=True`` = the cross-GPU negative
set (all-gather of the embeddings over RCCL, positives offset by rank*N, reduce-scatter of the
gathered copies' gradients; an extension — the reference never gathers, lines 55-56)."""
import torch
import torch.distributed as dist
from torch.autograd import Function

from ..core.sync_utils import collectives_active
from ..hip import ops


def _world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


class _NTXentFn(Function):
    @staticmethod
    def forward(ctx, h1, h2, T, co2_weight, gather):
        ctx.set_materialize_grads(False)        # (see loss/moco.py)
        h1, h2 = h1.contiguous(), h2.contiguous()
        B = h1.shape[0]
        coll = bool(gather) and collectives_active()
        ws = _world() if coll else 1
        if coll:
            a_all = torch.empty(ws * B, h1.shape[1], dtype=h1.dtype, device=h1.device)
            b_all = torch.empty_like(a_all)
            dist.all_gather_into_tensor(a_all, h1)
            dist.all_gather_into_tensor(b_all, h2)
            roff = dist.get_rank() * B
        else:
            a_all, b_all, roff = h1, h2, 0
        out, rowstats = ops.ntxent_fwd(h1, h2, a_all, b_all, roff, T, co2_weight)
        ctx.save_for_backward(h1, h2, a_all, b_all, rowstats)
        ctx.T, ctx.w, ctx.roff, ctx.ws, ctx.coll = T, co2_weight, roff, ws, coll
        loss, acc1 = out[0:1], out[1:2]
        ctx.mark_non_differentiable(acc1)
        return loss, acc1

    @staticmethod
    def backward(ctx, gloss, _gacc):
        h1, h2, a_all, b_all, rowstats = ctx.saved_tensors
        if gloss is None:
            return None, None, None, None, None
        da, db, dA, dB = ops.ntxent_bwd(h1, h2, a_all, b_all, rowstats, gloss.contiguous().float(),
                                        ctx.roff, ctx.T, ctx.w)
        B = h1.shape[0]
        if ctx.coll:
            ra, rb = torch.empty_like(da), torch.empty_like(db)
            dist.reduce_scatter_tensor(ra, dA)
            dist.reduce_scatter_tensor(rb, dB)
            da += ra
            db += rb
        elif da.is_cuda:
            # (library adds: an ATen `+=` would be a foreign launch inside a step that a native plan replays)
            ops.add_into(da, dA[ctx.roff:ctx.roff + B])
            ops.add_into(db, dB[ctx.roff:ctx.roff + B])
        else:
            da += dA[ctx.roff:ctx.roff + B]
            db += dB[ctx.roff:ctx.roff + B]
        return da, db, None, None, None


def nt_xent(h1, h2, T, co2_weight=3.0, gather=False):
    return _NTXentFn.apply(h1, h2, float(T), float(co2_weight), bool(gather))
