"""N>1 path on CPU: two gloo processes exercise the GradReducer (bucketed, order-safe,
overlappable all-reduce over the flat gradient buffer), grad_sync / param_sync, the key
all-gather used by the queue update, and the Trainer's replica initialisation."""
import os
import socket
from types import SimpleNamespace

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, fn, ret):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        ret[rank] = fn(rank, world)
    finally:
        dist.destroy_process_group()


def _spawn(fn, world=2):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), fn, ret), nprocs=world, join=True)
    return dict(ret)


def _fake_arena(sizes):
    off, slices = 0, []
    for n in sizes:
        slices.append((off, n))
        off += (n + 7) // 8 * 8
    return SimpleNamespace(grads=torch.zeros(off), param_slices=slices, reducer=None)


def _reducer_case(rank, world):
    from passl_amd.core.sync_utils import GradReducer
    sizes = [40, 8, 8, 100, 16, 16, 200, 8, 64, 24]
    arena = _fake_arena(sizes)
    opt = SimpleNamespace(grad_scale=1.0)
    red = GradReducer(arena, opt, bucket_elems=128)
    assert opt.grad_scale == 1.0 / world
    assert len(red.buckets) >= 3
    covered = sorted(i for b in red.buckets for i in b[2])
    assert covered == list(range(len(sizes)))                 # every parameter in exactly one bucket
    results = []
    for step in range(2):
        arena.grads.zero_()
        red.begin()
        # "backward": last layer first, but ranks report readiness in different interleavings
        order = list(range(len(sizes) - 1, -1, -1))
        if rank == 1:
            order[0], order[1] = order[1], order[0]
            order[4], order[6] = order[6], order[4]
        for i in order:
            off, n = arena.param_slices[i]
            arena.grads[off:off + n] = float((rank + 1) * (i + 1) + step)
            if i != 3 or step == 0:      # parameter 3 gets no gradient in step 1 (finish() covers it)
                red.mark_ready(i)
        red.finish()
        results.append(arena.grads.clone())
    return results


def test_grad_reducer_bucketed_allreduce():
    out = _spawn(_reducer_case)
    sizes = [40, 8, 8, 100, 16, 16, 200, 8, 64, 24]
    arena = _fake_arena(sizes)
    for step in range(2):
        assert torch.equal(out[0][step], out[1][step])
        for i, (off, n) in enumerate(arena.param_slices):
            expect = sum((r + 1) * (i + 1) + step for r in range(2))
            assert torch.all(out[0][step][off:off + n] == expect), (step, i)


def _sync_case(rank, world):
    from passl_amd.core.sync_utils import grad_sync, param_sync
    from passl_amd.modeling.architectures.moco import concat_all_gather
    torch.manual_seed(rank)
    lin = torch.nn.Linear(4, 3)
    lin.register_buffer('buf', torch.full((2,), float(rank)))
    param_sync(lin, src_rank=0)
    w_after = lin.weight.detach().clone()
    for p in lin.parameters():
        p.grad = torch.full_like(p, float(rank + 1))
    grad_sync([{'params': list(lin.parameters())}])
    keys = torch.full((2, 3), float(rank))
    gathered = concat_all_gather(keys)
    return w_after, lin.buf.clone(), lin.weight.grad.clone(), gathered


def test_param_sync_grad_sync_and_key_gather():
    out = _spawn(_sync_case)
    assert torch.equal(out[0][0], out[1][0])                # weights broadcast from rank 0
    assert torch.all(out[1][1] == 0)                        # buffers too
    assert torch.all(out[0][2] == 1.5) and torch.all(out[1][2] == 1.5)   # mean of (1, 2)
    for r in (0, 1):
        assert out[r][3].shape == (4, 3)
        assert torch.all(out[r][3][:2] == 0) and torch.all(out[r][3][2:] == 1)


def _trainer_case(rank, world):
    """Two Trainer replicas on CPU with a host-only model whose parameters live in an arena-like
    flat buffer: checks rank-dependent seeding + flat broadcast + reducer wiring."""
    from passl_amd.hip import config as hip_config
    hip_config.set_device('cpu')
    from passl_amd.hip.nn import EncoderArena, Linear
    from passl_amd.core.sync_utils import GradReducer, param_sync
    torch.manual_seed(100 + rank)
    enc = torch.nn.Sequential(Linear(8, 16), Linear(16, 8))
    for m in enc:
        torch.nn.init.normal_(m.weight)
    arena = EncoderArena(enc, trainable=True, dtype=torch.float32)
    holder = SimpleNamespace(arena_q=arena, parameters=lambda: enc.parameters(),
                             buffers=lambda: enc.buffers())
    before = arena.flat.clone()
    param_sync(holder, src_rank=0)
    red = GradReducer(arena, None, bucket_elems=64)
    red.begin()
    arena.grads.fill_(float(rank + 1))
    for i in range(len(arena.param_slices) - 1, -1, -1):
        arena.grad_ready([i])
    red.finish()
    return before, arena.flat.clone(), arena.grads.clone(), [p.grad.sum().item() for p in enc.parameters()]


def test_arena_replicas_and_reducer_via_arena_hooks():
    out = _spawn(_trainer_case)
    assert not torch.equal(out[0][0], out[1][0])             # different seeds before the sync
    assert torch.equal(out[0][1], out[1][1]) and torch.equal(out[0][1], out[0][0])
    assert torch.all(out[0][2] == 3.0) and torch.all(out[1][2] == 3.0)
    assert out[0][3] == out[1][3] and out[0][3][0] == 3.0 * 8 * 16   # p.grad views see the reduced buffer


def _simclr_gather_case(rank, world):
    """SimCLRContrastiveHead(multi_rank=True) on two gloo ranks: the collective pattern
    (all-gather embeddings, positives offset by rank*N, reduce-scatter of the column-role
    gradients) with the HIP kernels replaced by a torch restatement of their contract, compared
    with ONE process evaluating the loss of all 2N rows (mean over ranks of per-rank losses)."""
    from passl_amd.hip import ops
    from passl_amd.modeling.heads.simclr_contrastive_head import SimCLRContrastiveHead

    def head_terms(h1, h2, A, B_, roff, T, w):
        n, bl = h1.shape[0], A.shape[0]
        pos = torch.zeros(n, bl, dtype=torch.bool)
        pos[torch.arange(n), roff + torch.arange(n)] = True
        ninf = float('-inf')
        aa, ab, ba, bb = h1 @ A.t() / T, h1 @ B_.t() / T, h2 @ A.t() / T, h2 @ B_.t() / T
        aam, bbm = aa.masked_fill(pos, ninf), bb.masked_fill(pos, ninf)
        abm, bam = ab.masked_fill(pos, ninf), ba.masked_fill(pos, ninf)
        ce = (torch.logsumexp(torch.cat([ab, aam], 1), 1) - ab[pos]) + \
            (torch.logsumexp(torch.cat([ba, bbm], 1), 1) - ba[pos])
        x, y = torch.cat([aam, abm], 1), torch.cat([bam, bbm], 1)
        la, lb = torch.log_softmax(x, 1), torch.log_softmax(y, 1)
        fin = torch.isfinite(x)
        z = torch.zeros_like(la)
        kl = (lb.exp().detach() * (torch.where(fin, lb, z).detach() - torch.where(fin, la, z))).sum() + \
            (la.exp().detach() * (torch.where(fin, la, z).detach() - torch.where(fin, lb, z))).sum()
        return ce.mean() + w * kl / n

    saved = {}

    def fake_fwd(a, b, a_all, b_all, roff, T, w=3.0):
        loss = head_terms(a, b, a_all, b_all, roff, T, w)
        saved['args'] = (roff, T, w)
        return torch.stack([loss.detach(), torch.zeros(())]), torch.zeros(a.shape[0], 8)

    def fake_bwd(a, b, a_all, b_all, rowstats, gscale, roff, T, w=3.0):
        with torch.enable_grad():           # autograd.Function.backward runs under no_grad
            leaves = [t.detach().clone().requires_grad_(True) for t in (a, b, a_all, b_all)]
            (head_terms(*leaves, roff, T, w) * gscale.reshape(())).backward()
        return tuple(l.grad for l in leaves)

    ops.ntxent_fwd, ops.ntxent_bwd = fake_fwd, fake_bwd
    gen = torch.Generator().manual_seed(5)
    N, T = 6, 0.2
    full1 = torch.nn.functional.normalize(torch.randn(world * N, 128, generator=gen), dim=1)
    full2 = torch.nn.functional.normalize(torch.randn(world * N, 128, generator=gen), dim=1)
    h1 = full1[rank * N:(rank + 1) * N].clone().requires_grad_(True)
    h2 = full2[rank * N:(rank + 1) * N].clone().requires_grad_(True)
    head = SimCLRContrastiveHead(temperature=T, multi_rank=True)
    out = head(h1, h2)
    out['loss'].backward()
    assert saved['args'][0] == rank * N
    # single-process reference: DP averages the per-rank losses; every rank's loss depends on all rows
    f1 = full1.clone().requires_grad_(True)
    f2 = full2.clone().requires_grad_(True)
    tot = sum(head_terms(f1[r * N:(r + 1) * N], f2[r * N:(r + 1) * N], f1, f2, r * N, T, 3.0)
              for r in range(world))
    tot.backward()
    # this rank's gradient = d(sum of all ranks' losses)/d(own rows)  (reduce-scatter sums the
    # column-role parts contributed by the other ranks)
    e1 = float((h1.grad - f1.grad[rank * N:(rank + 1) * N]).abs().max())
    e2 = float((h2.grad - f2.grad[rank * N:(rank + 1) * N]).abs().max())
    return e1, e2, float(out['loss'].detach())


def test_simclr_head_cross_rank_gather():
    ret = _spawn(_simclr_gather_case)
    for rank in (0, 1):
        e1, e2, loss = ret[rank]
        assert e1 < 1e-5 and e2 < 1e-5, (rank, e1, e2)
        assert loss > 0


def _clip_gather_case(rank, world):
    """CLIPWrapper(multi_rank=True)'s collective pattern on two gloo ranks — all-gather of both
    modalities' normalised features, labels arange(N) + N*rank, reduce-scatter of the gathered copies'
    gradients — with the HIP kernels replaced by torch restatements of their contracts, against ONE
    process that evaluates every rank's loss on the full 2N x 2N logit matrix (BASELINE configs[4];
    pattern of passl/models/mocov3.py:187-198)."""
    import math
    import torch.nn.functional as F
    from passl_amd.hip import ops
    from passl_amd.modeling.backbones.clip import _CrossRankLogitsFn
    from passl_amd.modeling.heads.clip_head import CLIPHead

    def l2norm_fwd(x, eps=0.0):
        n = x.norm(dim=1).clamp_min(eps)
        return x / n[:, None], n

    def l2norm_bwd(dy, y, norm, dtype):
        return (dy - y * (dy * y).sum(1, keepdim=True)) / norm[:, None]

    def clip_scale(s, lo=-4.6, hi=4.6):
        a = s.detach().exp().clone()
        s.clamp_(lo, hi)
        return a

    def ce_fwd(scores, labels):
        lse = torch.logsumexp(scores, 1)
        loss = (lse - scores[torch.arange(scores.shape[0]), labels]).mean()
        return torch.stack([loss, torch.zeros(()), torch.zeros(())]), lse

    def ce_bwd(scores, lse, labels, g):
        p = (scores - lse[:, None]).exp()
        p[torch.arange(scores.shape[0]), labels] -= 1.0
        return p * g.reshape(()) / scores.shape[0]

    def dot_acc(a, b, out):
        out += (a * b).sum()

    ops.l2norm_fwd, ops.l2norm_bwd, ops.clip_scale = l2norm_fwd, l2norm_bwd, clip_scale
    ops.gemm_f32_nt = lambda a, b, alpha: alpha * (a @ b.t())
    ops.gemm_f32_gx = lambda g, x, alpha, trans=False: alpha * ((g.t() if trans else g) @ x)
    ops.dot_acc, ops.softmax_ce_fwd, ops.softmax_ce_bwd = dot_acc, ce_fwd, ce_bwd
    ops.add_into = lambda dst, src: dst.add_(src)

    gen = torch.Generator().manual_seed(9)
    N, D = 5, 32
    full_i = torch.randn(world * N, D, generator=gen)
    full_t = torch.randn(world * N, D, generator=gen)
    s0 = 1.3
    img = full_i[rank * N:(rank + 1) * N].clone().requires_grad_(True)
    txt = full_t[rank * N:(rank + 1) * N].clone().requires_grad_(True)
    scale = torch.tensor([s0], requires_grad=True)
    li, lt = _CrossRankLogitsFn.apply(img, txt, scale)
    assert li.shape == (N, world * N) and lt.shape == (N, world * N)
    labels = torch.arange(N) + N * rank
    out = CLIPHead()(li, lt, labels, labels)
    out['loss'].backward()
    ds = scale.grad.clone()
    dist.all_reduce(ds)                                      # sum over ranks of d loss_r / d s
    # single-process reference on the full matrices: sum over ranks of loss_r
    fi = full_i.clone().requires_grad_(True)
    ft = full_t.clone().requires_grad_(True)
    sr = torch.tensor([s0], requires_grad=True)
    L = sr.exp() * F.normalize(fi, dim=1) @ F.normalize(ft, dim=1).t()
    total, mine = 0.0, None
    for r in range(world):
        rows = slice(r * N, (r + 1) * N)
        lab = torch.arange(N) + N * r
        lr = F.cross_entropy(L[rows], lab) + F.cross_entropy(L.t()[rows], lab)
        total = total + lr
        if r == rank:
            mine = float(lr)
    total.backward()
    rows = slice(rank * N, (rank + 1) * N)
    e_loss = abs(float(out['loss']) - mine)
    e_img = float((img.grad - fi.grad[rows]).abs().max())
    e_txt = float((txt.grad - ft.grad[rows]).abs().max())
    e_s = abs(float(ds) - float(sr.grad))
    return e_loss, e_img, e_txt, e_s, abs(float(scale.detach()) - s0)


def test_clip_cross_rank_infonce():
    ret = _spawn(_clip_gather_case)
    for rank in (0, 1):
        e_loss, e_img, e_txt, e_s, e_clip = ret[rank]
        assert e_loss < 1e-5 and e_img < 1e-5 and e_txt < 1e-5 and e_s < 1e-4, (rank, ret[rank])
        assert e_clip < 1e-7            # 1.3 is inside the clip range: unchanged


def _sync_bn_math_case(rank, world):
    """SyncBatchNorm's exchange (ops._gather_doubles + the rank-ordered Chan combination of csrc/bn.hip, restated
    here in torch): every rank contributes {mean, M2, n} of its own batch; the combined moments equal those of the
    concatenated batch, and the backward sums gathered the same way give the joint batch's input gradient."""
    from passl_amd.hip.ops import _gather_doubles
    gen = torch.Generator().manual_seed(5)
    xs = [torch.randn(6 + r, 16, generator=gen, dtype=torch.float64) * (r + 1) + r for r in range(world)]
    x = xs[rank]
    mom = torch.stack([x.mean(0), ((x - x.mean(0)) ** 2).sum(0), torch.full((16,), float(x.shape[0]), dtype=torch.float64)])
    allm = _gather_doubles(mom)
    assert allm.shape == (world, 3, 16)
    n, mu, m2 = torch.zeros(16, dtype=torch.float64), torch.zeros(16, dtype=torch.float64), torch.zeros(16, dtype=torch.float64)
    for r in range(world):                     # bn_finalize_moments_kernel
        mb, qb, nb = allm[r]
        nn_, d = n + nb, mb - mu
        m2 = m2 + qb + d * d * n * nb / nn_
        mu = mu + d * nb / nn_
        n = nn_
    joint = torch.cat(xs)
    assert torch.allclose(mu, joint.mean(0), atol=1e-12) and torch.allclose(m2 / n, joint.var(0, unbiased=False), atol=1e-12)
    return float(mu.sum())


def test_sync_batchnorm_moment_exchange():
    out = _spawn(_sync_bn_math_case)
    assert out[0] == out[1]                    # rank-ordered combination: identical on every rank


def _eval_gather_case(rank, world):
    """ClassificationEvaluationLoop over two ranks: 21 samples, batch 4 per rank -> 3 batches per rank = 24 rows with
    3 repeats in the last global batch (DistributedBatchSampler pads by wrapping around); scores and labels of every
    batch are gathered and the repeats dropped, so the metric is over exactly the 21 samples
    (classification_loop.py:196-222)."""
    import torch.nn.functional as F
    from passl_amd.engine.loops import ClassificationEvaluationLoop
    total, bs = 21, 4
    gen = torch.Generator().manual_seed(11)
    xs, ys = torch.randn(total, 6, generator=gen), torch.randint(0, 5, (total,), generator=gen)
    W = torch.randn(6, 5, generator=gen)
    # global batch b = rows [b*8, b*8+8): rank r takes [b*8 + r*4, b*8 + r*4 + 4), indices wrap around at the end
    idx = [[(b * bs * world + rank * bs + i) % total for i in range(bs)] for b in range(3)]

    class Loader(list):
        dataset = range(total)

    loader = Loader([[xs[i], ys[i]] for i in idx])
    model = torch.nn.Linear(6, 5, bias=False)
    with torch.no_grad():
        model.weight.copy_(W.t())

    def loss_func(out, label):
        v = F.cross_entropy(out, label)
        return {'CELoss': v, 'loss': v}

    def metric_func(out, label):
        a = (out.argmax(1) == label).float().mean()
        return {'top1': a, 'metric': a}
    tr = SimpleNamespace(model=model, eval_loss_func=loss_func, eval_metric_func=metric_func, eval_dataloader=loader,
                         mode='eval', validating=True, cur_epoch_id=0)
    res = ClassificationEvaluationLoop(tr).run()
    want = float(((xs @ W).argmax(1) == ys).float().mean())
    local = torch.cat([xs[i] for i in idx]), torch.cat([ys[i] for i in idx])
    return res['top1'], want, res['loss'], float(F.cross_entropy(local[0] @ W, local[1]))


def test_evaluation_loop_gathers_scores_and_drops_the_sampler_repeats():
    out = _spawn(_eval_gather_case)
    for rank in (0, 1):
        top1, want, loss, local_loss = out[rank]
        assert abs(top1 - want) < 1e-6, (rank, top1, want)              # the metric is global and exact
        assert abs(loss - local_loss) < 1e-6                           # the loss average is per rank (as the reference)
    assert out[0][0] == out[1][0]


def _mae_loop_case(rank, world):
    """engine/loops/mae_pretrain_loop.py under data parallelism, on a CPU stand-in for the model (the loop is device
    agnostic): parameters broadcast once, ONE reducer, armed before the last micro-batch of every accumulation window."""
    from passl_amd.engine.loops import mae_pretrain_loop as loop
    torch.manual_seed(100 + rank)                                     # replicas start DIFFERENT: param_sync must fix it
    n = 24
    flat = torch.randn(n)
    grads = torch.zeros(n)

    class Toy(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(flat)                         # the parameter IS the flat buffer
            self.w.grad = grads
            self.arena = SimpleNamespace(flat=self.w.data, grads=grads, param_slices=[(0, 8), (8, 16)], reducer=None)

        def forward(self, x, mask_ratio=0.75):
            return ((self.w * x).sum() ** 2) * 1e-2, None, None

    class SGD(object):
        def __init__(self, model):
            self.m, self.lr, self.grad_scale = model, 0.0, 1.0

        def set_lr(self, lr):
            self.lr = lr

        def step(self):
            if self.m.arena.reducer is not None:
                self.m.arena.reducer.finish()
            with torch.no_grad():
                self.m.w -= self.lr * self.grad_scale * self.m.w.grad

        def clear_grad(self):
            self.m.w.grad.zero_()

    model = Toy()
    opt = SGD(model)
    data = [torch.full((n,), float(1 + rank + 2 * i)) for i in range(4)]          # each rank its own shard
    args = SimpleNamespace(accum_iter=2, mask_ratio=0.75, lr=0.05, min_lr=0.0, warmup_epochs=0, epochs=2, print_freq=2)
    begins = []
    orig_begin = loop.GradReducer.begin
    loop.GradReducer.begin = lambda self: (begins.append(1), orig_begin(self))[1]
    try:
        for epoch in range(2):
            loop.train_one_epoch(model, data, opt, epoch, args, log=lambda *_: None)
    finally:
        loop.GradReducer.begin = orig_begin
    assert len(begins) == 4                                           # 2 epochs x 2 accumulation windows
    assert model._passl_grad_reducer is model.arena.reducer and opt.grad_scale == 1.0 / world
    return model.w.detach().clone()


def test_mae_pretrain_loop_averages_gradients_across_ranks():
    out = _spawn(_mae_loop_case)
    assert torch.equal(out[0], out[1])                                # replicas stay identical
    # ... and equal to one process doing both ranks' batches with averaged gradients from rank 0's start
    torch.manual_seed(100)
    w = torch.randn(24)
    lr_at = lambda e: 0.05 * 0.5 * (1.0 + __import__('math').cos(__import__('math').pi * e / 2))
    for epoch in range(2):
        for win in range(2):
            lr = lr_at(epoch + (2 * win) / 4)
            g = torch.zeros(24)
            for rank in range(2):
                for i in (2 * win, 2 * win + 1):
                    x = torch.full((24,), float(1 + rank + 2 * i))
                    g += (2e-2 * (w * x).sum() * x) / 2              # loss / accum_iter
            w = w - lr * g / 2                                        # mean over the two ranks
    assert torch.allclose(out[0], w, rtol=1e-5, atol=1e-6), (out[0], w)


def _wire_case(rank, world):
    """GradReducer(wire_dtype=bf16): every bucket travels as bf16 (half the bytes per link), the buffer stays fp32."""
    from passl_amd.core.sync_utils import GradReducer
    sizes = [40, 8, 8, 100, 16, 16, 200, 8, 64, 24]
    out = {}
    for name, wire in (('fp32', None), ('bf16', torch.bfloat16)):
        arena = _fake_arena(sizes)
        red = GradReducer(arena, SimpleNamespace(grad_scale=1.0), bucket_elems=128, wire_dtype=wire)
        assert (red.wire is not None) == (wire is not None)
        torch.manual_seed(7 + rank)
        red.begin()
        for i in range(len(sizes) - 1, -1, -1):
            off, n = arena.param_slices[i]
            arena.grads[off:off + n] = torch.randn(n) * (10.0 ** (i % 4 - 2))      # four decades of magnitudes
        out['own'] = arena.grads.clone()
        for i in range(len(sizes) - 1, -1, -1):
            red.mark_ready(i)
        red.finish()
        assert arena.grads.dtype == torch.float32
        out[name] = arena.grads.clone()
    return out


def test_grad_reducer_bf16_wire_matches_fp32_to_bf16_rounding():
    out = _spawn(_wire_case)
    for name in ('fp32', 'bf16'):
        assert torch.equal(out[0][name], out[1][name])                 # both ranks hold the same sums
    ref, got = out[0]['fp32'], out[0]['bf16']
    assert not torch.equal(ref, got)                                   # it really went through bf16 ...
    # ... each rank's contribution is rounded to bf16 (2^-9 relative to ITS magnitude) and so is their sum
    bound = 2.0 ** -8 * (out[0]['own'].abs() + out[1]['own'].abs() + ref.abs()) + 1e-30
    assert bool(((got - ref).abs() <= bound).all()), float(((got - ref).abs() / bound).max())


def test_default_buckets_are_one_big_bucket_and_a_small_tail(monkeypatch):
    """Round 6: by default the gradient buffer is all-reduced as TWO buckets — the head of the buffer (the first layers,
    whose gradients arrive last: <= 1.5 M elements and <= 1/8 of the buffer) and everything else — instead of 4 x 28 MB:
    only the last bucket's collective is exposed, and every bucket is a live host call between two plan segments."""
    from passl_amd.core.sync_utils import GradReducer
    monkeypatch.delenv('PASSL_DP_BUCKETS', raising=False)
    # R50-like: stem + 4 stages + projector (elements per parameter tensor, flat-buffer order)
    sizes = [9408, 64, 64] + [70000] * 3 + [400000] * 3 + [2300000] * 3 + [5000000] * 3 + [4194304, 2048, 262144, 128]
    arena = _fake_arena(sizes)
    red = GradReducer(arena, SimpleNamespace(grad_scale=1.0))
    assert len(red.buckets) == 2
    (s0, e0, i0), (s1, e1, i1) = red.buckets              # launch order: the big one first
    assert e0 == arena.param_slices[-1][0] + sizes[-1] and s1 == 0 and e1 == s0
    assert sorted(i0 + i1) == list(range(len(sizes)))
    assert e1 - s1 <= 3 * 512 * 1024 + max(sizes[:9]) and (e1 - s1) * 8 <= sum(sizes) + 8 * len(sizes) + max(sizes[:9]) * 8
    assert len(i1) >= 6                                    # stem + the first stages: what backward finishes with
    # tiny buffers: one bucket would do, two are still valid
    small = GradReducer(_fake_arena([40, 8, 8, 100]), SimpleNamespace(grad_scale=1.0))
    assert sorted(i for b in small.buckets for i in b[2]) == [0, 1, 2, 3] and 1 <= len(small.buckets) <= 2
    monkeypatch.setenv('PASSL_DP_BUCKETS', '4')
    assert len(GradReducer(_fake_arena(sizes), SimpleNamespace(grad_scale=1.0)).buckets) >= 3
