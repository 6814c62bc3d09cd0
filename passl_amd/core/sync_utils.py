"""Data-parallel gradient / parameter synchronisation over RCCL (backend "nccl" on ROCm) or gloo.

What it replaces in the reference:
  * v110: ``fleet.distributed_model(model)`` = paddle.DataParallel — bucketed gradient all-reduce
    (sum, then / nranks) overlapped with backward, parameters broadcast from rank 0 at wrap time
    (passl_v110/engine/trainer.py:172-183, 218-219)
  * v2:   ``grad_sync`` / ``param_sync`` (passl/core/sync_utils.py:18-69): blocking per-parameter
    all_reduce after backward + scale by 1/nranks; broadcast of params and buffers from rank 0

MI355X design: gradients already live in ONE flat fp32 buffer (EncoderArena.grads), so a bucket is
a contiguous slice — no flatten/unflatten copies.  The backward kernels' host code reports each
layer's parameters as ready (arena.grad_ready); when every parameter of a bucket is ready the
slice is all-reduced asynchronously: the collective is ISSUED from the weight-gradient (side) stream after it has been
made to wait for the bucket's other producers (hip/streams.py:comm_stream), RCCL runs it on its stream, the main chain
waits for nothing until the optimizer — it overlaps with the remaining backward kernels.  Buckets are walked from
the END of the buffer (the last layers finish first in backward).  The 1/world_size scale is folded into
the optimizer kernel (no extra pass).  Bucket layout (round 6): TWO buckets — everything but
the head of the buffer (R50: 106 MB, complete 60 % into backward, fully hidden) and the <= 6 MB head, whose
all-reduce is the only one backward cannot hide; every further bucket is a live host call inside the step
(+0.18 ms each, measured).  xGMI is point-to-point (7 links/GPU): one 106 MB collective is as
bandwidth-bound as a ring gets.  The communicator itself is created at the first collective, not at
init_process_group (engine/trainer.py: its existence from init on costs the step 8 %).
"""
import os
import time

import torch
import torch.distributed as dist

# PASSL_DP_DRYRUN=1 (1-rank diagnostic): keep every plan cut, stream edge and host call of the data-parallel path but skip
# the torch.distributed calls themselves — separates what the MACHINERY costs from what the collective library costs
_DRYRUN = os.environ.get('PASSL_DP_DRYRUN') == '1'
# PASSL_DP_DIAG=noedges,nogather,noreducer (1-rank diagnostics of the same kind: drop one part of the machinery)
_DIAG = set(filter(None, os.environ.get('PASSL_DP_DIAG', '').split(',')))


class _Issuer(object):
    """PASSL_DP_THREAD=1: the live collective calls of a step are made by ONE helper thread instead of the thread that
    drives the step.  Every stream edge of a bucket is already enqueued when its call is posted (hip/streams.py:
    gather_into), so the call's only job is to hand the bucket to the collective library — if that call holds the host
    for a while (communicator bookkeeping, a launch queue that is full), the step's own launches no longer wait behind
    it.  Order: one thread, FIFO — identical on every rank; ``drain`` (before the handles are waited for) returns when
    every posted call has been made."""

    def __init__(self, device):
        import queue
        import threading
        self.q = queue.Queue()
        self.err = None
        self.device = device

        def run():
            if device is not None and device.type == 'cuda':
                torch.cuda.set_device(device)
            while True:
                fn = self.q.get()
                try:
                    if fn is None:
                        return
                    if self.err is None:
                        fn()
                except BaseException as e:            # surfaced by drain() on the step's thread
                    self.err = e
                finally:
                    self.q.task_done()
        self.t = threading.Thread(target=run, name='passl-dp-issuer', daemon=True)
        self.t.start()

    def post(self, fn):
        self.q.put(fn)

    def drain(self):
        self.q.join()
        if self.err is not None:
            e, self.err = self.err, None
            raise e


_issuers = {}


def _issuer(device):
    if os.environ.get('PASSL_DP_THREAD') != '1':
        return None
    key = (device.type, device.index)
    if key not in _issuers:
        _issuers[key] = _Issuer(device)
    return _issuers[key]


def _ws(group=None):
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def dp_forced():
    """PASSL_DP_FORCE=1: issue every data-parallel collective even in a 1-rank job.  A world-size-1
    RCCL communicator still creates the transport, launches the collective kernels on its stream and
    orders them against the compute stream — this is how the single-GPU test box loads and runs RCCL
    (tests/test_dp_gpu.py::test_rccl_world1); results must equal the collective-free run bit for bit."""
    return os.environ.get('PASSL_DP_FORCE') == '1'


def collectives_active(group=None):
    """True when the DP collectives of this process must be issued."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or dp_forced()


def _cast(src, dst):
    """dst[...] = src[...] across fp32 <-> bf16 (contiguous slices of equal length): the library's cast kernels on the
    device (launches a step plan can replay), torch on the host (gloo tests)."""
    if src.is_cuda:
        from ..hip import lib as L
        fn = L.load().passl_hip_cast_f32_to_bf16 if src.dtype == torch.float32 else L.load().passl_hip_cast_bf16_to_f32
        L.check(fn(L.ptr(src), L.ptr(dst), src.numel(), L.stream()), 'gradient wire cast')
    else:
        dst.copy_(src)


class GradReducer(object):
    """Bucketed, overlapped all-reduce over a flat gradient buffer."""

    def __init__(self, arena, optimizer=None, bucket_elems=None, group=None, wire_dtype=None):
        """bucket_elems: fp32 gradients per bucket; ``PASSL_DP_BUCKETS=n`` asks for n equal buckets (bench.py
        --dp-buckets); default: two buckets, a big one and a small tail (see below).
        wire_dtype: torch.bfloat16 (or ``PASSL_DP_WIRE=bf16``) sends every bucket as bf16 — half the bytes per xGMI
        link (56 MB instead of 112 MB for R50; the last bucket, which backward cannot hide, shrinks with it): the
        bucket is cast into a resident bf16 twin of the gradient buffer right before its all-reduce and cast back
        after the final wait, so accumulation, the 1 / world scale and the optimizer stay fp32.  The SUM over ranks is
        then formed in bf16 by the collective: gradients agree with the fp32 wire to bf16 rounding (~3 significant
        digits), not bit for bit — opt-in; the default (fp32) is the reference's behaviour."""
        tail_elems = None
        n_train = getattr(arena, 'n_train', None) or sum(n for _o, n in arena.param_slices)
        if bucket_elems is None:
            n_b = int(os.environ.get('PASSL_DP_BUCKETS', '0') or 0)
            if n_b > 0:
                bucket_elems = -(-n_train // n_b)
            else:
                # default (round 6): TWO buckets — everything but the head of the buffer in one, the head (the first
                # layers' parameters, whose gradients arrive last: at most 1.5 M elements = 6 MB, and at most 1/8 of the
                # buffer) in a second one.  Only the LAST bucket's all-reduce cannot hide under backward whatever the
                # bucket count, so it should be small; every further bucket costs a live call on the host between two
                # segments of the step (measured with a world-1 communicator, profiles/r06_dp_overhead.txt: +0.18 ms of
                # step time per bucket — 4 x 28 MB: +0.70 ms, 1 bucket: +0.16 ms).  The big bucket (R50: layer3, layer4
                # and the projector, 106 MB) is complete 60 % into the backward pass; the tail is stem + layer1 + layer2.
                bucket_elems = 1 << 62
                tail_elems = int(os.environ.get('PASSL_DP_TAIL_ELEMS', str(3 * 512 * 1024)))
                tail_elems = min(tail_elems, n_train // 8)
        if wire_dtype is None and os.environ.get('PASSL_DP_WIRE', '').lower() in ('bf16', 'bfloat16'):
            wire_dtype = torch.bfloat16
        if wire_dtype not in (None, torch.float32, torch.bfloat16):
            raise ValueError('wire_dtype must be None / torch.float32 / torch.bfloat16')
        self.wire = None if wire_dtype in (None, torch.float32) else \
            torch.empty(arena.grads.numel(), dtype=wire_dtype, device=arena.grads.device)
        self.arena = arena
        self.group = group
        self.world = _ws(group)
        self._forced = collectives_active(group)
        self.grads = arena.grads
        slices = arena.param_slices
        # buckets = runs of consecutive parameters, built from the end of the buffer
        self.buckets = []          # (start_elem, end_elem, [param indices])
        cur, cur_n, end = [], 0, None
        for idx in range(len(slices) - 1, -1, -1):
            off, n = slices[idx]
            if end is None:
                end = off + n
            cur.append(idx)
            cur_n += n
            # (tail scheme: the big bucket closes at the first parameter that starts inside the head of the buffer)
            if cur_n >= bucket_elems or idx == 0 or (tail_elems and not self.buckets and off <= tail_elems):
                self.buckets.append((off, end, cur))
                cur, cur_n, end = [], 0, None
        self.bucket_of = {}
        for b, (_s, _e, idxs) in enumerate(self.buckets):
            for i in idxs:
                self.bucket_of[i] = b
        self._pending = [0] * len(self.buckets)
        self._launched = [False] * len(self.buckets)
        self._ready = set()
        self._handles = []
        self._active = False
        self._home = None
        self.measure = False          # bench.py: time what the compute stream waits for in finish()
        self.host_ms, self.host_calls = 0.0, 0      # host time spent inside the live collective calls (bench.py reports it)
        self._exposed = []
        arena.reducer = self
        if optimizer is not None:
            optimizer.grad_scale = 1.0 / self.world

    def begin(self):
        """Call right before loss.backward()."""
        self._pending = [len(b[2]) for b in self.buckets]
        self._launched = [False] * len(self.buckets)
        self._ready = set()
        self._handles = []
        self._active = True
        self._home = torch.cuda.current_stream(self.grads.device) if self.grads.is_cuda else None

    def _launch(self, b):
        s, e, _ = self.buckets[b]
        self._launched[b] = True
        if not (self.world > 1 or self._forced):
            return          # nothing to order: the end-of-backward join of hip/streams.py covers whoever reads .grad
        from ..hip.replay import host_call
        grads, group = self.grads, self.group
        comm = None
        if self.grads.is_cuda:
            # The bucket's producers sit on up to three streams: weight gradients on the side stream
            # (hip/streams.py), BatchNorm / bias gradients on the stream backward() was called on, and this very
            # call may come from a node autograd runs on the side stream (a forked downsample branch).  The
            # collective is ordered behind the CURRENT stream only, so it is issued from a stream that waits for all of
            # them (streams.comm_stream: the side stream itself) — the main chain itself waits for nothing here (until
            # round 5 it joined the side stream in front of every bucket: four waits per backward pass on the
            # critical chain).  The waits go through hip/streams.py: a recorded step replays them.
            from ..hip import streams
            comm = streams.comm_stream(self.grads.device)
            if 'noedges' not in _DIAG:
                streams.gather_into(comm, self.grads.device, extra=(self._home,) if self._home is not None else ())
        if self.wire is not None:
            # a launch of the library, OUTSIDE the host call: a recorded step replays it as part of the plan (on the
            # issuing stream), the live collective below then reads the twin
            if comm is not None:
                with torch.cuda.stream(comm):
                    _cast(self.grads[s:e], self.wire[s:e])
            else:
                _cast(self.grads[s:e], self.wire[s:e])
            grads = self.wire

        def collective():
            # a live call also when the step is replayed from a native plan (hip/replay.py: the plan is cut here);
            # the replayed segment in front of it contains the issuing stream's waits for the producers
            t0 = time.perf_counter()
            if _DRYRUN:
                h = None                 # (diagnostic, world 1 only: the cuts and stream edges without the library call)
            elif comm is not None:
                with torch.cuda.stream(comm):
                    h = dist.all_reduce(grads[s:e], op=dist.ReduceOp.SUM, group=group, async_op=True)
            else:
                h = dist.all_reduce(grads[s:e], op=dist.ReduceOp.SUM, group=group, async_op=True)
            if h is not None:
                self._handles.append(h)
            self.host_ms += 1e3 * (time.perf_counter() - t0)
            self.host_calls += 1
        iss = _issuer(self.grads.device)
        host_call(collective if iss is None else (lambda: iss.post(collective)))

    def mark_ready(self, index):
        if not self._active or index in self._ready:
            return
        self._ready.add(index)
        b = self.bucket_of[index]
        self._pending[b] -= 1
        # launch in bucket order so that every rank issues the collectives in the same sequence
        while True:
            nxt = next((i for i, l in enumerate(self._launched) if not l), None)
            if nxt is None or self._pending[nxt] > 0:
                break
            self._launch(nxt)

    def finish(self):
        """Call before the optimizer reads the gradients: launches what is left (parameters that
        received no gradient this step) and waits for every collective."""
        if not self._active:
            return
        for b in range(len(self.buckets)):
            if not self._launched[b]:
                self._launch(b)
        from ..hip.replay import host_call

        def wait_all():
            h0 = time.perf_counter()
            iss = _issuer(self.grads.device)
            if iss is not None:
                iss.drain()                    # every bucket's call has been made: the handle list is complete
            timed = self.measure and self.grads.is_cuda and self._handles
            if timed:
                e0 = torch.cuda.Event(enable_timing=True)
                e0.record()
            for h in self._handles:
                h.wait()
            self.host_ms += 1e3 * (time.perf_counter() - h0)
            self.host_calls += 1
            if timed:
                e1 = torch.cuda.Event(enable_timing=True)
                e1.record()
                self._exposed.append((e0, e1))
            self._handles = []
        if self._handles or self.world > 1 or self._forced:
            host_call(wait_all)                 # (live at every replay of a recorded step: see _launch)
            if self.wire is not None:
                for s, e, _ in self.buckets:    # the reduced sums back into the fp32 buffer the optimizer reads
                    _cast(self.wire[s:e], self.grads[s:e])
        self._active = False
        uses = getattr(self.arena, '_uses', None)
        if uses:
            uses.clear()                        # (a forward whose backward never ran must not hold the next step back)

    def exposed_ms(self):
        """Average time per step that the compute stream spent between "backward finished" and "every gradient
        collective finished" over the finish() calls made with ``measure`` on: the all-reduce time that backward
        did NOT hide (the last bucket can only start when backward ends).  Synchronises."""
        if not self._exposed:
            return None
        torch.cuda.synchronize()
        ms = [a.elapsed_time(b) for a, b in self._exposed]
        self._exposed = []
        return sum(ms) / len(ms)


class ReducerGroup(object):
    """One overlapped GradReducer per trainable arena behind the single-reducer interface the loops / hooks use
    (``begin`` before the backward pass; every optimizer waits for its own arena's reducer in ``step``).  A model whose
    parameter groups live in separate arenas (SimSiam: encoder + predictor, passl/models/simsiam.py with the recipe's
    two learning rates) used to fall back to the loop's blocking per-buffer ``grad_sync`` after backward; with a group
    its gradient buckets are all-reduced from inside the backward pass like everyone else's.  The collectives are
    issued in the order the buckets become ready, which is a property of the graph: identical on every rank."""

    def __init__(self, arenas, optimizer=None, group=None):
        self.reducers = [GradReducer(a, None, group=group) for a in arenas]
        self.world = self.reducers[0].world
        if optimizer is not None:
            optimizer.grad_scale = 1.0 / self.world

    @property
    def buckets(self):
        return [b for r in self.reducers for b in r.buckets]

    @property
    def grads(self):
        return torch.cat([r.grads for r in self.reducers]) if len(self.reducers) > 1 else self.reducers[0].grads

    @property
    def measure(self):
        return self.reducers[0].measure

    @measure.setter
    def measure(self, v):
        for r in self.reducers:
            r.measure = v

    def begin(self):
        for r in self.reducers:
            r.begin()

    def finish(self):
        for r in self.reducers:
            r.finish()

    def exposed_ms(self):
        vals = [v for v in (r.exposed_ms() for r in self.reducers) if v is not None]
        return sum(vals) if vals else None


@torch.no_grad()
def grad_sync(param_groups, comm_group=None, grad_avg=True):
    """v2 spelling (passl/core/sync_utils.py:18-43): blocking all_reduce of every parameter's
    gradient (+ average).  Arena-backed parameters are reduced as ONE flat call per arena."""
    nranks = _ws(comm_group)
    if not collectives_active(comm_group):
        return
    seen = []
    for group in param_groups:
        for p in group['params']:
            if p.grad is None:
                continue
            a = getattr(p, '_passl_arena', None)
            if a is not None:
                if a not in seen:
                    seen.append(a)
                    dist.all_reduce(a.grads, group=comm_group)
                    if grad_avg:
                        a.grads.mul_(1.0 / nranks)
                continue
            dist.all_reduce(p.grad, group=comm_group)
            if grad_avg:
                p.grad.mul_(1.0 / nranks)


@torch.no_grad()
def param_sync(model, src_rank=0, comm_group=None):
    """Broadcast parameters and buffers from ``src_rank`` (passl/core/sync_utils.py:46-69).
    Arena-backed state is one broadcast per flat buffer."""
    if not collectives_active(comm_group):
        return
    arenas = [getattr(model, n) for n in ('arena_q', 'arena_k', 'arena') if getattr(model, n, None) is not None]
    flat_ptrs = set()
    for a in arenas:
        dist.broadcast(a.flat, src=src_rank, group=comm_group)
        flat_ptrs.add(a.flat.untyped_storage().data_ptr())
    seen = set()
    for t in list(model.parameters()) + list(model.buffers()):
        if t.untyped_storage().data_ptr() in flat_ptrs or id(t) in seen:
            continue
        seen.add(id(t))
        dist.broadcast(t, src=src_rank, group=comm_group)
    if hasattr(model, 'sync_runtime_state') and next(model.parameters()).is_cuda:
        model.sync_runtime_state()
