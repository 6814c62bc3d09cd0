"""The v2 MAE pre-training loop — reference tasks/ssl/mae/engine_pretrain.py:30-107 (``train_one_epoch``) with the
schedule of tasks/ssl/mae/util/lr_sched.py:24-33 (linear warm-up, then half-cycle cosine, set per ITERATION from the
fractional epoch).  The MAE task does not go through ``passl.engine.Engine``: its script drives
``model(samples, mask_ratio)`` -> ``loss`` -> scaled backward -> AdamW directly, with gradient accumulation over
``accum_iter`` iterations.

Differences by design: bf16 compute needs no loss scaler (``loss_scaler(loss, optimizer, update_grad=...)`` becomes
backward + step); the loss stays a device tensor — the reference reads ``loss.item()`` and synchronises the device
every iteration (engine_pretrain.py:70,91) — and is converted once per ``print_freq`` window; a non-finite loss is
detected at that point (the reference exits on the spot).

Data parallelism: the reference wraps the model in ``paddle.DataParallel`` (tasks/ssl/mae/main_pretrain.py), which
broadcasts the parameters once and averages the gradients of every backward pass.  Here, when a process group is up,
``train_one_epoch`` does the same through the flat arena: ``param_sync`` once per model, one overlapped ``GradReducer``
over the arena's gradient buffer, armed right before the backward pass of the LAST micro-batch of an accumulation
window (the buckets are all-reduced from inside that backward pass; ``optimizer.step()`` waits for them)."""
import math

import torch

from ...core.sync_utils import GradReducer, collectives_active, param_sync
from ...hip import ops


def adjust_learning_rate(optimizer, epoch, args):
    """Decay the learning rate with half-cycle cosine after warmup (util/lr_sched.py:24-33)."""
    if epoch < args.warmup_epochs:
        lr = args.lr * epoch / args.warmup_epochs
    else:
        lr = args.min_lr + (args.lr - args.min_lr) * 0.5 * \
            (1. + math.cos(math.pi * (epoch - args.warmup_epochs) / (args.epochs - args.warmup_epochs)))
    optimizer.set_lr(lr)
    return lr


def _data_parallel(model, optimizer):
    """The model's gradient reducer under data parallelism (created, with the one-time parameter broadcast, at the first
    call); None in a single-process run."""
    if not collectives_active():
        return None
    reducer = getattr(model, '_passl_grad_reducer', None)
    if reducer is None:
        arena = getattr(model, 'arena', None)
        if arena is None:
            raise RuntimeError('MAE pre-training under data parallelism needs a model whose parameters live in a flat '
                               'arena (passl.models.build_model): without one the ranks would train divergent replicas')
        param_sync(model)
        reducer = GradReducer(arena, optimizer)
        model._passl_grad_reducer = reducer
    return reducer


def train_one_epoch(model, data_loader, optimizer, epoch, args, log=print):
    """args: accum_iter, mask_ratio, lr, min_lr, warmup_epochs, epochs, print_freq, max_train_step (optional).
    -> {'loss': mean over the epoch, 'lr': last learning rate}."""
    model.train()
    accum_iter = int(getattr(args, 'accum_iter', 1))
    print_freq = int(getattr(args, 'print_freq', 20))
    max_train_step = getattr(args, 'max_train_step', None)
    optimizer.clear_grad()
    pending, total, count, lr = [], 0.0, 0, None
    n_iter = len(data_loader)
    reducer = _data_parallel(model, optimizer)

    def flush():
        nonlocal total, count
        if pending:
            vals = torch.stack([p.detach().reshape(()).float() for p in pending]).cpu()
            if not bool(torch.isfinite(vals).all()):
                raise FloatingPointError('Loss is {}, stopping training'.format(vals.tolist()))
            total += float(vals.sum())
            count += len(pending)
            del pending[:]

    for data_iter_step, batch in enumerate(data_loader):
        samples = batch[0] if isinstance(batch, (list, tuple)) else batch
        global_iter_step = data_iter_step + n_iter * epoch
        if max_train_step is not None and global_iter_step >= max_train_step:
            log('step({}) >= max_train_step({}), training stops early.'.format(global_iter_step, max_train_step))
            break
        # a per iteration (instead of per epoch) lr scheduler
        if data_iter_step % accum_iter == 0:
            lr = adjust_learning_rate(optimizer, data_iter_step / n_iter + epoch, args)
        loss, _, _ = model(samples, mask_ratio=args.mask_ratio)
        pending.append(loss)
        loss = loss / accum_iter if accum_iter != 1 else loss
        if reducer is not None and (data_iter_step + 1) % accum_iter == 0:
            reducer.begin()                 # the gradients of the whole window are complete after THIS backward pass
        loss.backward(ops.ones_like_cached(loss) if loss.is_cuda else None)
        if (data_iter_step + 1) % accum_iter == 0:
            optimizer.step()
            optimizer.clear_grad()
        if (data_iter_step + 1) % print_freq == 0 or data_iter_step + 1 == n_iter:
            flush()
            log('Epoch: [{}]  [{}/{}]  lr: {:.6f}  loss: {:.4f}'.format(epoch, data_iter_step + 1, n_iter, lr,
                                                                         total / max(count, 1)))
    flush()
    return {'loss': total / max(count, 1), 'lr': lr}
