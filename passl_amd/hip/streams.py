"""Second HIP stream for work that is independent of the main chain.

A training step is one long dependency chain on the current stream (conv -> BN -> conv ... ->
data-gradient -> BN backward -> data-gradient ...) whose kernels alternate between MFMA/latency-bound
(implicit GEMMs at 20-25 % matrix-pipe utilisation) and HBM-bound (BatchNorm passes).  One kind of
launch hangs off that chain without feeding it:

  * weight gradients: dW of a layer needs (x, dy) and is consumed by the optimizer only.

(MoCo's key path is NOT hoisted: its EMA covers the BatchNorm running statistics the query forward has
just updated, moco.py:82-90 — a true dependency.)  Side work is issued on ONE side stream per device,
so that its workgroups fill the CUs next to the main chain's (different bottlenecks share a CU:
LDS-heavy GEMM workgroups + register-light streaming workgroups) instead of extending the chain.  Ordering: the side stream waits for an event recorded on
the main stream at the hand-off point (its inputs are complete), tensors it reads are
``record_stream``-ed (the caching allocator must not recycle them early), and the main stream waits
for the side stream before anything consumes the results (``join``): at the end of every backward
pass (autograd engine callback) and before a gradient bucket is all-reduced.  Every kernel stays
deterministic; only the interleaving changes.
"""
import contextlib

import torch

from . import config

_streams = {}
_dirty = {}


def side_stream(device):
    key = device.index if device.index is not None else torch.cuda.current_device()
    s = _streams.get(key)
    if s is None:
        s = _streams[key] = torch.cuda.Stream(device=device)
    return s


def enabled(t):
    return config.overlap() and t.is_cuda


@contextlib.contextmanager
def on_side(device, reads=(), in_backward=False):
    """Run the enclosed launches on the side stream, after everything enqueued on the current stream
    so far.  ``reads``: tensors allocated on the main stream that the enclosed kernels read."""
    main = torch.cuda.current_stream(device)
    s = side_stream(device)
    s.wait_event(main.record_event())
    with torch.cuda.stream(s):
        yield s
    for t in reads:
        if t is not None:
            t.record_stream(s)
    _dirty[s.device.index] = True
    if in_backward:
        # join at the end of this backward pass: whoever reads .grad afterwards sees finished work
        # (one callback per hand-off; all but the first find nothing left to wait for)
        torch.autograd.Variable._execution_engine.queue_callback(lambda: join(device))


def join(device):
    """Make the current stream wait for everything issued on the side stream so far."""
    if not device.type == 'cuda':
        return
    s = side_stream(device)
    key = s.device.index
    if _dirty.get(key):
        torch.cuda.current_stream(device).wait_stream(s)
        _dirty[key] = False
