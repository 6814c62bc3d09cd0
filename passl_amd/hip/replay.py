"""The training step as a NATIVE LAUNCH PLAN: recorded once, replayed without Python in the launch path.

The reference's iteration (passl_v110/engine/trainer.py:287-337: model forward, then OptimizerHook: clear_grad ->
backward -> step, hooks/optimizer_hook.py:25-50) is ~1 400 kernel launches here.  Issued one by one from Python they
cost the host 14-19 ms per MoCo step (24-26 ms of GPU time) and ALL of a CLIP ViT-B/32 step; round 3 measured that a
HIP graph does not help on this ROCm (hipGraphLaunch: ~11 us of host time per kernel node, and a capture cannot hold
the forked branches whose backward autograd runs on another stream).  The library makes every launch itself, so it
can simply write the launches down (csrc/plan.h / plan.hip) while ONE step executes normally:

  * kernels: handle, grid, block, LDS bytes, stream, argument bytes — whatever dispatch decision the library took;
  * cross-stream edges: every event record / stream wait of the product path goes through hip/streams.py
    (`record_event`, `wait_event`, `wait_stream`), which reports them to the recording plan; the edges autograd adds
    on its own around a node that runs off the main stream are replaced by a conservative equivalent
    (`streams.autograd_node_entry`);
  * segments: a host callback (`StepPlan.host_call`) closes the current segment and is re-run between two segments at
    every replay — that is where a torch.distributed collective stays a live call.

Replay = `passl_hip_plan_replay(plan, segment)`: a C loop over hipLaunchKernel / hipEventRecord /
hipStreamWaitEvent on the recorded streams.  What must hold for that to mean the same step again:

  * memory: every allocation of the recorded step comes from a private pool of torch's caching allocator
    (`_cuda_beginAllocateToPool`: all threads, i.e. the autograd thread too) that nobody else allocates from afterwards
    — recorded addresses stay valid, and block reuse inside the step is the same stream-ordered reuse as when it ran;
  * step-varying scalars live in device memory: learning rate / beta^t (`_DeviceHyper.push_hyper`, a live H2D copy
    before every replay), MoCo's queue pointer (advanced by the enqueue kernel);
  * the batch is copied into the recorded input tensors unless the caller passes the very same resident tensors;
  * no foreign launch inside the step: a TorchDispatchMode watches the recording and refuses a plan (eager steps
    continue, with a warning) if an ATen op that launches a kernel ran inside it.  hip/ops.py has library kernels for
    the fills / copies / casts a step needs.

Opt-in per model (``graph_safe``: no random numbers drawn inside the step, every step-varying value on the device).
``PASSL_PLAN=0`` is the kill switch.  Replayed and eager steps are bit-identical
(tests/test_moco_gpu.py::test_step_plan_replay_is_bit_identical).
"""
import ctypes as C
import logging
import os
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

from . import lib as L
from . import streams

logger = logging.getLogger('passl')


def plans_enabled():
    return os.environ.get('PASSL_PLAN', '1') != '0'


# ATen ops that run no device kernel (views, metadata, allocation); everything else seen while a plan records is a
# launch the plan would not contain
_NO_KERNEL_PREFIXES = (
    'aten.view', 'aten._unsafe_view', 'aten.reshape', 'aten._reshape_alias', 'aten.as_strided', 'aten.slice',
    'aten.select', 'aten.detach', 'aten.alias', 'aten.permute', 'aten.transpose', 'aten.t.', 'aten.expand',
    'aten.unsqueeze', 'aten.squeeze', 'aten.empty', 'aten.empty_like', 'aten.empty_strided', 'aten.new_empty',
    'aten.is_', 'aten.sym_', 'aten.size', 'aten.stride', 'aten.numel', 'aten.dim', 'aten.unbind', 'aten.split',
    'aten.chunk', 'aten.narrow', 'aten.lift_fresh', 'aten.view_as', 'aten.flatten', 'aten.unflatten',
    'aten.record_stream', 'aten._has_compatible_shallow_copy_type', 'aten.set_', 'aten.resize_',
    'aten.result_type', 'aten.can_cast', 'aten.storage_offset', 'aten.movedim', 'aten.swapaxes', 'prim.',
    'aten.contiguous')


class _ForeignOpWatch(TorchDispatchMode):
    """Collects the ATen ops dispatched while a plan records that (may) launch device kernels."""

    def __init__(self):
        super().__init__()
        self.foreign = {}
        self.paused = 0                 # > 0 inside a host call: those launches are live at every replay

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        out = func(*args, **(kwargs or {}))
        if not self.paused and not name.startswith(_NO_KERNEL_PREFIXES):
            on_device = any(torch.is_tensor(a) and a.is_cuda for a in args) or \
                (torch.is_tensor(out) and out.is_cuda)
            if name.startswith('aten.clone') and torch.is_tensor(out) and out.numel() == 0:
                on_device = False
            if on_device:
                site = '?'
                for fr in reversed(traceback.extract_stack()[:-1]):
                    if 'passl_amd' in fr.filename and not fr.filename.endswith('replay.py'):
                        site = '%s:%d' % (os.path.basename(fr.filename), fr.lineno)
                        break
                key = (name, site)
                self.foreign[key] = self.foreign.get(key, 0) + 1
        return out


class _Recorder(object):
    """What hip/streams.py talks to while a plan records."""

    _serial = 0

    def __init__(self, lib, handle, device, watch=None):
        self.watch = watch
        _Recorder._serial += 1
        self.serial = _Recorder._serial
        self.lib, self.handle, self.device = lib, handle, device
        self.main = torch.cuda.current_stream(device).cuda_stream
        self.dropped_waits = 0
        self.node_syncs = 0
        self.host_calls = []            # [(segment index that FOLLOWS the call, callable)]

    def on_record(self, stream, ev):
        idx = self.lib.passl_hip_plan_event_record(self.handle, stream.cuda_stream)
        if idx < 0:
            L.check(idx, 'plan_event_record')
        ev._passl_plan_event = (self.serial, idx)

    def on_wait(self, stream, ev):
        tag = getattr(ev, '_passl_plan_event', None)
        if tag is None or tag[0] != self.serial:
            # an event of an earlier step: everything it covers was joined into the stream the step starts on before
            # this step began (end-of-backward join, optimizer), and replays are enqueued in step order
            self.dropped_waits += 1
            return
        L.check(self.lib.passl_hip_plan_stream_wait(self.handle, stream.cuda_stream, tag[1]), 'plan_stream_wait')

    def on_backward_node(self, device):
        cur = L.stream()
        if cur == self.main:
            return
        idx = self.lib.passl_hip_plan_event_record(self.handle, self.main)
        if idx < 0:
            L.check(idx, 'plan_event_record')
        L.check(self.lib.passl_hip_plan_stream_wait(self.handle, cur, idx), 'plan_stream_wait')
        self.node_syncs += 1

    def cut(self, fn):
        seg = self.lib.passl_hip_plan_cut(self.handle)
        if seg < 0:
            L.check(seg, 'plan_cut')
        self.host_calls.append((seg, fn))


def host_call(fn):
    """Run ``fn()`` now; if a step plan is recording, also close its current segment here and re-run ``fn()`` at this
    point of every replay.  For work inside a step that is not a launch of this library and must stay a live call: a
    torch.distributed collective (its tensors are the recorded ones: same addresses at every replay)."""
    rec = streams._recorder
    if rec is None:
        return fn()
    rec.cut(fn)
    watch = rec.watch
    if watch is not None:
        watch.paused += 1
    try:
        return fn()
    finally:
        if watch is not None:
            watch.paused -= 1


class StepPlan(object):
    def __init__(self, fn, optimizers=(), replay_hooks=(), warmup=3, enabled=True, strict=None):
        """fn(*tensors) -> dict: the COMPLETE step (forward, clear_grad, backward, optimizer step).
        optimizers: objects with ``push_hyper()`` (their step scalars are refreshed before every replay).
        replay_hooks: callables run after every replay (host mirrors of device state, e.g. MoCo's `_ptr`).
        strict: refuse the plan when a foreign (ATen) launch was seen while recording (default: PASSL_PLAN_STRICT != 0)."""
        self.fn = fn
        self.optimizers = list(optimizers)
        self.replay_hooks = list(replay_hooks)
        self.warmup = int(warmup)
        self.enabled = bool(enabled)
        self.strict = (os.environ.get('PASSL_PLAN_STRICT', '1') != '0') if strict is None else bool(strict)
        self.calls = 0
        self.handle = None
        self.failed = None              # why recording was refused (eager steps from then on)
        self.static_in = None
        self.static_out = None
        self._src = None
        self._pool = None
        self._host_calls = []
        self._main = None
        self.replays = 0
        self.eager_fallbacks = 0        # steps of another shape than the recorded one (run eagerly)
        self._warned_shape = False
        self._scratch_pins = None       # the workspace buffers whose addresses the recorded launches carry
        self.info = {}
        self.foreign = {}

    # same vocabulary as hip/graph.py:StepGraph
    @property
    def captured(self):
        return self.handle is not None

    def _fits(self, data):
        """The batch has the recorded step's shapes and dtypes (a recorded step is shape-specialised)."""
        if len(data) != len(self.static_in):
            return False
        for d, s in zip(data, self.static_in):
            if torch.is_tensor(d) != torch.is_tensor(s):
                return False
            if torch.is_tensor(d) and (tuple(d.shape) != tuple(s.shape) or d.dtype != s.dtype):
                return False
        return True

    def _load_inputs(self, data):
        from . import ops
        for i, (d, s) in enumerate(zip(data, self.static_in)):
            if not torch.is_tensor(d):
                continue
            src = self._src[i]
            if src is not None and src[0] is d and src[1] == d._version:
                continue                        # the same resident tensor object, unmodified (see hip/graph.py)
            if d.is_cuda and d.is_contiguous() and s.is_contiguous():
                ops.copy_into(s, d)
            else:
                s.copy_(d, non_blocking=True)
            self._src[i] = (d, d._version)

    def _outputs(self):
        from . import ops
        return {k: (ops.clone(v.detach()) if torch.is_tensor(v) and v.is_cuda else v)
                for k, v in self.static_out.items()}

    def _record(self, data):
        lib = L.load()
        if not (hasattr(torch.cuda, 'MemPool') and hasattr(torch._C, '_cuda_beginAllocateToPool') and
                hasattr(torch._C, '_cuda_endAllocateToPool')):
            # (no private allocator pool on this torch build: recorded addresses could not be kept stable)
            self.failed = 'torch build without the caching-allocator pool API (torch.cuda.MemPool)'
            logger.warning('StepPlan: %s; continuing with eager launches', self.failed)
            return self.fn(*data)
        dev = next(d.device for d in data if torch.is_tensor(d))
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        torch.cuda.synchronize(dev)
        torch.cuda.empty_cache()                # the warm-up steps' cached blocks go back: the plan owns its pool
        self.static_in = [d.clone() if torch.is_tensor(d) else d for d in data]
        self._src = [(d, d._version) if torch.is_tensor(d) else None for d in data]
        for o in self.optimizers:
            o.push_hyper()                      # a live copy in front of the recorded step, never part of the plan
        streams.reset()
        torch.cuda.synchronize(dev)
        handle = C.c_void_p()
        L.check(lib.passl_hip_plan_create(C.byref(handle)), 'plan_create')
        pool = torch.cuda.MemPool()
        watch = _ForeignOpWatch()
        rec = _Recorder(lib, handle, dev, watch)
        began = False
        try:
            torch._C._cuda_beginAllocateToPool(idx, pool.id)        # every thread: the autograd thread allocates too
            try:
                L.check(lib.passl_hip_plan_record_begin(handle), 'plan_record_begin')
                began = True
                streams._recorder = rec
                with watch:
                    out = self.fn(*self.static_in)
            finally:
                streams._recorder = None
                if began:
                    rc = lib.passl_hip_plan_record_end(handle)
                torch._C._cuda_endAllocateToPool(idx, pool.id)
                # begin took a reference of its own (torch.cuda.use_mem_pool pairs end with release): without this the
                # pool outlives its MemPool object and reset() gives nothing back — a second plan, or an eager step of a
                # large batch after a reset, then finds the whole recorded step still reserved
                if hasattr(torch._C, '_cuda_releasePool'):
                    torch._C._cuda_releasePool(idx, pool.id)
            L.check(rc, 'plan_record_end')
        except Exception:
            lib.passl_hip_plan_destroy(handle)
            raise
        streams.reset()
        if not isinstance(out, dict):
            lib.passl_hip_plan_destroy(handle)
            raise TypeError('StepPlan: the step function must return a dict of outputs')
        self.foreign = dict(watch.foreign)
        self.info = dict(segments=int(lib.passl_hip_plan_info(handle, 0)), kernels=int(lib.passl_hip_plan_info(handle, 1)),
                         event_records=int(lib.passl_hip_plan_info(handle, 2)),
                         stream_waits=int(lib.passl_hip_plan_info(handle, 3)),
                         memsets=int(lib.passl_hip_plan_info(handle, 4)),
                         arg_bytes=int(lib.passl_hip_plan_info(handle, 5)),
                         streams=int(lib.passl_hip_plan_info(handle, 7)),
                         dropped_waits=rec.dropped_waits, node_syncs=rec.node_syncs, host_calls=len(rec.host_calls))
        if self.foreign and self.strict:
            # the step that just ran was a correct eager step; only its REPLAY would miss these launches
            lib.passl_hip_plan_destroy(handle)
            self.failed = 'ATen launches inside the step: ' + ', '.join(
                '%s x%d (%s)' % (n, c, s) for (n, s), c in sorted(self.foreign.items()))
            logger.warning('StepPlan: not replaying this step (%s); continuing with eager launches', self.failed)
            self.static_in = self._src = None
            del pool
            return out
        self.static_out = out
        self.handle = handle
        self._pool = pool                       # keeps the recorded step's addresses reserved
        self._host_calls = list(rec.host_calls)
        self._main = rec.main                   # the stream the step was called on (replays must be, too)
        # The recorded launches carry raw pointers into the shared grow-only scratch (ops.workspace: weight-gradient
        # slabs, InfoNCE dq slabs, column-sum partials), which was allocated from the general pool during the warm-up
        # steps.  workspace.get() REPLACES a buffer when a later eager call asks for more (validation, another model,
        # a tail batch through the eager fallback): without these references the old buffer would be freed and every
        # replay would write its slabs into memory that belongs to somebody else.
        from . import ops
        self._scratch_pins = ops.workspace.pin()
        return None

    def run(self, *data):
        if not self.enabled or not plans_enabled() or self.failed is not None:
            return self.fn(*data)
        if self.handle is None:
            if self.calls < self.warmup:
                self.calls += 1
                return self.fn(*data)
            out = self._record(data)            # (the step executed while it was recorded: host mirrors are advanced)
            if out is not None:
                return out                      # recording refused: this was a plain eager step
            return self._outputs()
        if L.stream() != self._main:
            # the live pieces around a replay (input copies, the learning-rate push, output clones, host calls) are
            # ordered against the recorded launches through the CURRENT stream: only valid on the recorded one
            return self.fn(*data)
        if not self._fits(data):
            # another batch shape (the tail batch of a `drop_last: False` loader, reference
            # configs/simclr/simclr_r50_IM.yaml:88-90): this one step runs eagerly, the plan stays for the next
            if not self._warned_shape:
                self._warned_shape = True
                logger.warning('StepPlan: a batch of another shape / dtype than the recorded step (%s): eager launches '
                               'for such steps, replays continue for the recorded shape',
                               [tuple(d.shape) for d in data if torch.is_tensor(d)])
            self.eager_fallbacks += 1
            return self.fn(*data)
        self._load_inputs(data)
        for o in self.optimizers:
            o.push_hyper()
        lib = L.load()
        seg = 0
        for nxt, fn in self._host_calls:
            while seg < nxt:
                L.check(lib.passl_hip_plan_replay(self.handle, seg), 'plan_replay')
                seg += 1
            fn()
        while seg < self.info['segments']:
            L.check(lib.passl_hip_plan_replay(self.handle, seg), 'plan_replay')
            seg += 1
        for o in self.optimizers:
            o._hyper_pushed = False             # consumed by the replayed update kernel
        for h in self.replay_hooks:
            h()
        self.replays += 1
        return self._outputs()

    def reset(self):
        """Drop the recorded plan (shapes / model structure changed): the next calls warm up and record again."""
        if self.handle is not None:
            torch.cuda.synchronize()
            L.load().passl_hip_plan_destroy(self.handle)
        self.handle = self.static_in = self.static_out = self._src = self._pool = self._scratch_pins = None
        if torch.cuda.is_available():
            torch.cuda.empty_cache()            # the plan's pool: its blocks are ordinary cached blocks now
        self._host_calls = []
        self.failed = None
        self.calls = 0

    def __del__(self):
        try:
            if self.handle is not None:
                torch.cuda.synchronize()
                L.load().passl_hip_plan_destroy(self.handle)
                self.handle = None
        except Exception:
            pass
