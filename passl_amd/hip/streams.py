"""Second HIP stream for work that is independent of the main chain.

A training step is one long dependency chain on the current stream (conv -> BN -> conv ... ->
data-gradient -> BN backward -> data-gradient ...) whose kernels alternate between MFMA/latency-bound
(implicit GEMMs at 20-25 % matrix-pipe utilisation) and HBM-bound (BatchNorm passes).  One kind of
launch hangs off that chain without feeding it:

  * weight gradients: dW of a layer needs (x, dy) and is consumed by the optimizer only.

(MoCo's key path is not hoisted as a whole: its EMA covers the BatchNorm running statistics the query forward has
just updated, moco.py:82-90 — a true dependency, but a LAYER-WISE one: the key encoder runs on a stream of its own
one trunk stage behind the query forward, see key_stream and architectures/moco.py:_train_iter_overlapped.)  Side work is issued on ONE side stream per device,
so that its workgroups fill the CUs next to the main chain's (different bottlenecks share a CU:
LDS-heavy GEMM workgroups + register-light streaming workgroups) instead of extending the chain.  Ordering: the side stream waits for an event recorded on
the main stream at the hand-off point (its inputs are complete), and the main stream waits
for the side stream before anything consumes the results (``join``): at the end of every backward
pass (autograd engine callback) and before the optimizer's update kernel; a gradient bucket's all-reduce is ordered
behind it through ``comm_stream`` / ``gather_into`` without making the main stream wait (round 6).  Every kernel stays
deterministic; only the interleaving changes.

Memory.  The tensors a hand-off reads (allocated on the main stream) must not be recycled while the
side stream still reads them.  ``Tensor.record_stream`` would do that, but the host enqueues a whole
backward pass long before the GPU executes it, so a recorded block is never seen as finished when
the allocator looks and NOTHING the side stream touched is reused inside a step: at SimCLR batch
512 the reserved pool grew from 177 GB to the 288 GB of the device and the allocator started to
free and re-allocate (measured 3-10x slower steps).  Instead a hand-off keeps REFERENCES to those
tensors until the held bytes exceed a budget (8 % of the device memory: every hand-off of a batch-256
MoCo step fits — a lag of 2 / 8 / 16 layers measured 9541 / 9691 / 9698 img/s, all of them 9817 —
while SimCLR at batch 512 retires after ~10 layers); when the oldest hand-off retires, the main
stream first waits (on the GPU) for its event — everything the main stream enqueues afterwards is
ordered behind the side stream's reads, so the plain stream-ordered reuse of the caching allocator is
safe again.  The opposite direction (tensors from the SIDE stream's pool read on the main stream: a
forked branch's output, its input gradient, a staged input) needs nothing: every piece of side-stream
work begins with a wait on a main-stream event recorded at its hand-off, i.e. after the main-stream
readers were enqueued.
"""
import collections
import contextlib
import os

import torch

from . import config

# ---- cross-stream edges, visible to a recording step plan (hip/replay.py) ------------------------------------
# Every event record / stream wait of the product path goes through these three functions.  They do what the torch
# calls do; while a step plan records they ALSO report the edge to it (a plan replays the library's launches on the
# recorded streams and must reproduce the ordering between them).
_recorder = None          # hip/replay.py sets it for the duration of a recording


def record_event(stream):
    ev = stream.record_event()
    if _recorder is not None:
        _recorder.on_record(stream, ev)
    return ev


def wait_event(stream, ev):
    stream.wait_event(ev)
    if _recorder is not None:
        _recorder.on_wait(stream, ev)


def wait_stream(stream, other):
    """stream.wait_stream(other) — which IS wait_event(other.record_event())."""
    wait_event(stream, record_event(other))


def autograd_node_entry(device):
    """Call at the top of a custom backward.  Autograd runs a node on the stream its forward ran on and orders that
    stream behind the producers of the node's incoming gradients with events of its own — edges a recording plan
    cannot see.  For a node that runs off the plan's main stream (the backward of a forked downsample branch) the
    plan gets a conservative equivalent: this stream waits for everything enqueued on the main stream so far (the
    producer ran there, or handed its result over through a GradSlot's own event)."""
    if _recorder is not None:
        _recorder.on_backward_node(device)


_streams = {}
_pending = {}       # device index -> deque of (event recorded on the side stream, tensors kept alive)
_HOLD_FRAC = float(os.environ.get('PASSL_OVERLAP_HOLD_FRAC', '0.08'))
_held_bytes = {}
_owners = {}
_budget = {}


def _aux_priority():
    """Priority of the side / fork / key streams (PASSL_AUX_PRIORITY, default 0 = the framework's default; HIP: lower
    number = served first).  The chain that bounds the step runs on the stream the step is called on; see
    ``main_stream`` for running THAT one at a higher priority."""
    return int(os.environ.get('PASSL_AUX_PRIORITY', '0'))


_main_streams = {}


def main_stream(device):
    """PASSL_MAIN_PRIORITY=p (unset: None): a stream of priority p (-1 = served before the default-priority side /
    key streams) for the step's dependency chain; the Trainer / bench run the iteration under it."""
    p = os.environ.get('PASSL_MAIN_PRIORITY')
    if p is None or p == '':
        return None
    key = device.index if device.index is not None else torch.cuda.current_device()
    s = _main_streams.get(key)
    if s is None:
        s = _main_streams[key] = torch.cuda.Stream(device=device, priority=int(p))
    return s


def side_stream(device):
    key = device.index if device.index is not None else torch.cuda.current_device()
    s = _streams.get(key)
    if s is None:
        s = _streams[key] = torch.cuda.Stream(device=device, priority=_aux_priority())
    return s


_fork_streams = {}


def fork_stream(device):
    """Stream of forked forward/backward branches (a bottleneck block's downsample branch): the
    weight-gradient stream.  A stream of its own (PASSL_FORK_OWN_STREAM=1), so that a branch's backward
    chain does not queue behind weight-gradient launches, measured no better (9871 vs 9913 img/s)."""
    if os.environ.get('PASSL_FORK_OWN_STREAM', '0') != '1':
        return side_stream(device)
    key = device.index if device.index is not None else torch.cuda.current_device()
    s = _fork_streams.get(key)
    if s is None:
        s = _fork_streams[key] = torch.cuda.Stream(device=device, priority=_aux_priority())
    return s


_key_streams = {}


def key_stream(device):
    """Stream of MoCo's key-encoder forward (architectures/moco.py:_KeyPipeline): it runs one trunk stage behind
    the query forward and must not queue behind the side stream's forked downsample branches."""
    key = device.index if device.index is not None else torch.cuda.current_device()
    s = _key_streams.get(key)
    if s is None:
        s = _key_streams[key] = torch.cuda.Stream(device=device, priority=_aux_priority())
    return s


_comm_streams = {}


def comm_stream(device):
    """Stream a gradient bucket's collective is ISSUED from (core/sync_utils.py:GradReducer).  torch.distributed
    orders its communication stream behind the CURRENT stream, and a bucket has producers on two streams (weight
    gradients on the side stream, BatchNorm / bias gradients on the stream backward() runs on).  Making the MAIN
    stream wait for the side stream in front of every bucket (rounds 1-5) put a cross-stream wait — and whatever the
    side stream was behind by — on the step's critical chain four times per backward pass.  Instead the issuing stream
    waits for the producers (``gather_into``) and the main chain waits for nobody until the optimizer.

    The issuing stream is the SIDE stream itself (the weight gradients' own stream: it is behind the main chain anyway,
    so its wait for the main stream's BatchNorm / bias gradients costs nothing).  A dedicated fifth stream — the first
    form of this round — measured +0.47 ms per forced data-parallel step against +0.09 ms this way
    (profiles/r06_dp_overhead.txt, call 32; 4 hardware queues by default: main, side, key and torch.distributed's own
    stream fill them, a fifth stream shares a queue with one of the product streams and its waits hold that queue).
    PASSL_DP_COMM_STREAM=own restores the dedicated stream."""
    key = device.index if device.index is not None else torch.cuda.current_device()
    s = _comm_streams.get(key)
    if s is None:
        if os.environ.get('PASSL_DP_COMM_STREAM', 'side') != 'own':
            s = side_stream(device)
        else:
            s = torch.cuda.Stream(device=device, priority=_aux_priority())
        _comm_streams[key] = s
    return s


def gather_into(target, device, extra=()):
    """Order ``target`` behind everything issued so far on the current stream, the side / fork / key streams and
    ``extra`` — without making any of THEM wait.  Queued side work is handed over first."""
    flush_side(device)
    key = device.index if device.index is not None else torch.cuda.current_device()
    seen = []
    for st in (torch.cuda.current_stream(device),) + tuple(extra) + tuple(
            t.get(key) for t in (_streams, _fork_streams, _key_streams)):
        if st is not None and st != target and st not in seen:
            seen.append(st)
            wait_stream(target, st)


_TIGHT_FRAC = float(os.environ.get('PASSL_OVERLAP_MAX_RESERVED_FRAC', '0.8'))
_tight = {}          # device index -> [calls, tight?, capacity]


def enabled(t):
    """Side-stream work for tensor t's device?  Off when the caching allocator already holds most of the
    device (a second stream means a second pool of cached blocks: +60 GB at SimCLR batch 512): the launches
    then stay on the current stream and draw from its pool.  The allocator is asked every 256th call only
    (memory_reserved() assembles the full statistics dictionary: ~0.1 ms of host time)."""
    if not (config.overlap() and t.is_cuda):
        return False
    key = t.device.index if t.device.index is not None else torch.cuda.current_device()
    st = _tight.get(key)
    if st is None:
        st = _tight[key] = [0, False, torch.cuda.get_device_properties(key).total_memory]
    if st[0] % 256 == 0:
        st[1] = torch.cuda.memory_reserved(key) >= _TIGHT_FRAC * st[2]
    st[0] += 1
    return not st[1]


@contextlib.contextmanager
def on_side(device, reads=(), in_backward=False):
    """Run the enclosed launches on the side stream, after everything enqueued on the current stream
    so far.  ``reads``: tensors allocated on the main stream that the enclosed kernels read."""
    main = torch.cuda.current_stream(device)
    s = side_stream(device)
    wait_event(s, record_event(main))
    with torch.cuda.stream(s):
        yield s
    key = s.device.index
    q = _pending.setdefault(key, collections.deque())
    held = tuple(t for t in reads if t is not None)
    nbytes = sum(t.numel() * t.element_size() for t in held)
    q.append((record_event(s), held, nbytes))
    owners = _owners.setdefault(key, [])
    if main != s and main not in owners:
        owners.append(main)         # streams that hand work over (the main stream, the fork stream)
    _held_bytes[key] = _held_bytes.get(key, 0) + nbytes
    budget = _budget.get(key)
    if budget is None:
        budget = _budget[key] = int(_HOLD_FRAC * torch.cuda.get_device_properties(key).total_memory)
    while _held_bytes[key] > budget and len(q) > 1:
        done, held, nbytes = q.popleft()
        # the held tensors may come from the pool of any handing-off stream (a forked branch reads the
        # main stream's block input): later work of ALL of them is ordered behind the side stream's reads
        for owner in owners:
            wait_event(owner, done)
        _held_bytes[key] -= nbytes
        del held
    if in_backward:
        _queue_end_join(device, key)


def _queue_end_join(device, key):
    if key not in _join_queued:
        # join at the end of this backward pass: whoever reads .grad afterwards sees finished work.  ONE callback per
        # backward pass: a join is an event record + a cross-stream wait, and on this GPU every such wait costs the
        # waiting stream ~10 us even when the event has long fired — one callback per hand-off (r02-r03) was 55
        # waits = 0.32 ms of idle GPU in front of MoCo's optimizer launch and 167 waits = 1.6 ms in front of MAE's
        # (profiles/r04_trace_timeline_*.txt)
        _join_queued.add(key)

        def _join_at_end():
            flush_side(device)               # (while the flag is still set: its hand-off must not queue another callback)
            _join_queued.discard(key)
            join(device)
        torch.autograd.Variable._execution_engine.queue_callback(_join_at_end)


_join_queued = set()     # devices with an end-of-backward join callback already queued for the running backward pass

# ---- side work of a backward pass, handed over in batches ---------------------------------------------------------
# Every hand-off makes the side stream wait for an event of the handing-off stream, and such a wait stalls the waiting
# stream for ~10 us on this GPU whether or not the event has fired.  A backward pass hands over 55
# (MoCo) to 167 (MAE) pieces of work — weight gradients, bias column sums, LayerNorm parameter folds — none of which
# anything but the optimizer / the gradient reducer reads.  They are therefore queued and handed over PASSL_SIDE_BATCH at
# a time (default 4; 1 = one hand-off per piece as before): one wait per batch.  ``join`` flushes the queue first, so
# whoever orders itself behind the side stream (end of backward, a gradient bucket, the optimizer) sees all of it.
_SIDE_BATCH = max(1, int(os.environ.get('PASSL_SIDE_BATCH', '4')))
_deferred = {}           # device index -> [(fn, tensors it reads, stream that produced them)]


def side_later(device, fn, reads=(), urgent=False):
    """Queue ``fn()`` (launches that only the optimizer / gradient reducer consume) for the side stream; call from a
    backward node.  ``reads``: tensors the launches read (kept alive until handed over, then by ``on_side``).
    ``urgent``: hand the queue over now (a long launch near the end of the backward pass)."""
    key = device.index if device.index is not None else torch.cuda.current_device()
    lst = _deferred.setdefault(key, [])
    lst.append((fn, tuple(t for t in reads if t is not None), torch.cuda.current_stream(device)))
    _queue_end_join(device, key)
    if urgent or len(lst) >= _SIDE_BATCH:
        flush_side(device)


def flush_side(device):
    key = device.index if device.index is not None else torch.cuda.current_device()
    lst = _deferred.get(key)
    if not lst:
        return
    _deferred[key] = []
    s = side_stream(device)
    cur = torch.cuda.current_stream(device)
    # on_side orders the side stream behind the CURRENT stream; pieces queued by a node that ran on another stream
    # (a forked branch's backward runs on the side stream itself, the rest on the main stream) need that stream's tail
    for st in {id(item[2]): item[2] for item in lst}.values():
        if st != s and st != cur:
            wait_event(s, record_event(st))
    with on_side(device, reads=tuple(t for _fn, r, _st in lst for t in r), in_backward=True):
        for fn, _r, _st in lst:
            fn()


def reset():
    """Forget every hand-off in flight and every stream that handed work over (call with the device idle).
    hip/graph.py calls it around a capture: a stream left in ``_owners`` by earlier EAGER steps (the default
    stream) must not be made to wait on an event recorded inside the capture — the wait would pull it into the
    capture, nothing would join it back, and ending the capture fails."""
    _pending.clear()
    _owners.clear()
    _held_bytes.clear()
    _join_queued.clear()
    _deferred.clear()


def join(device):
    """Make the current stream wait for everything issued so far on the side stream (and the fork stream, when it
    is a stream of its own).  Unconditional once such a stream exists: work can reach the side stream without
    passing through ``on_side`` — autograd runs the backward of a forked downsample branch there by stream
    affinity, BatchNorm gradients included — so a "dirty" flag kept by ``on_side`` would miss it (round-2 advisor
    finding).  A wait on an idle stream costs a few microseconds of host time."""
    if not device.type == 'cuda':
        return
    flush_side(device)                   # queued side work first: the caller orders itself behind ALL of it
    key = device.index if device.index is not None else torch.cuda.current_device()
    cur = torch.cuda.current_stream(device)
    waited = False
    for table in (_streams, _fork_streams, _key_streams):
        # (the key stream too: CLIP runs its text tower there, forward AND — by autograd's stream affinity — backward)
        s = table.get(key)
        if s is not None and s != cur:
            wait_stream(cur, s)
            waited = True
    q = _pending.get(key)
    if q:
        last = q[-1][0]
        for owner in _owners.get(key, ()):
            if owner != cur:
                wait_event(owner, last)
        q.clear()                   # every handing-off stream is ordered behind the side stream's reads
        _held_bytes[key] = 0
    return waited
