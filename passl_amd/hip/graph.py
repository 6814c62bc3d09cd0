"""The training step as ONE HIP graph.

The reference's iteration (passl_v110/engine/trainer.py:287-337: model forward, then OptimizerHook:
clear_grad -> backward -> step) is ~700-1500 kernel launches here, each issued from Python through ctypes: the
host needs 56-65 % of the GPU's time to enqueue a MoCo step and 95-100 % for CLIP ViT-B/32 (scratch/host_slack.py),
and on a data-parallel node one slow host stalls every rank at the next gradient bucket.  The C ABI was built
for capture — no allocation, no host synchronisation, launches on the caller's stream — so the whole step is
captured once (torch.cuda.graph: torch's allocator hands out a private pool, its streams and events become graph
edges, RCCL collectives are capturable) and replayed with one launch per step.

What changes from step to step is kept OUT of the graph's launch parameters:
  * learning rate, AdamW's beta^t: device scalars written before every replay (solver/optimizer.py:_DeviceHyper);
  * MoCo's queue pointer: the `queue_ptr` buffer on the device, advanced by the enqueue kernel;
  * the batch: copied into the captured input tensors (skipped when the caller passes the same, unmodified
    resident tensors again, as the synthetic loader does).
Outputs (loss, accuracies) are returned as fresh 1-element clones: hooks may keep them across steps.

``StepGraph.run`` executes eagerly for the first ``warmup`` calls (kernel attributes, plan caches, workspaces and
the allocator settle), captures on the next call and replays from then on.  ``PASSL_GRAPH=0`` (or
``enabled=False``) keeps every call eager; eager and replayed steps are bit-identical
(tests/test_moco_gpu.py::test_step_graph_replay_is_bit_identical).
"""
import os

import torch

from . import streams


def graphs_enabled():
    return os.environ.get('PASSL_GRAPH', '1') != '0'          # the kill switch; opting in is the Trainer's cfg.hip_graph


class StepGraph(object):
    def __init__(self, fn, optimizers=(), replay_hooks=(), warmup=3, enabled=True):
        """fn(*tensors) -> dict: the COMPLETE step (forward, clear_grad, backward, optimizer step).
        optimizers: objects with ``push_hyper()`` (their step scalars are refreshed before every replay).
        replay_hooks: callables run after every replay (host mirrors of device state, e.g. MoCo's `_ptr`)."""
        self.fn = fn
        self.optimizers = list(optimizers)
        self.replay_hooks = list(replay_hooks)
        self.warmup = int(warmup)
        self.enabled = bool(enabled)
        self.calls = 0
        self.graph = None
        self.static_in = None
        self.static_out = None
        self._src = None
        self.replays = 0

    @property
    def captured(self):
        return self.graph is not None

    def _load_inputs(self, data):
        for i, (d, s) in enumerate(zip(data, self.static_in)):
            if not torch.is_tensor(d):
                continue
            src = self._src[i]
            if src is not None and src[0] is d and src[1] == d._version:
                # the SAME tensor object (held by a strong reference, so its address cannot have been recycled for
                # another batch) and nobody wrote to it since: the resident synthetic batch.  (An address / version /
                # shape key is not an identity: a fresh batch from a real loader often lands on the freed address of
                # the previous one with version 0 — round-3 advisor finding.)
                continue
            if tuple(d.shape) != tuple(s.shape) or d.dtype != s.dtype:
                raise RuntimeError('StepGraph: input %d changed shape / dtype (%s %s -> %s %s); a captured step is '
                                   'shape-specialised' % (i, tuple(s.shape), s.dtype, tuple(d.shape), d.dtype))
            s.copy_(d, non_blocking=True)
            self._src[i] = (d, d._version)

    def _outputs(self):
        return {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in self.static_out.items()}

    def _capture(self, data):
        dev = next(d.device for d in data if torch.is_tensor(d))
        torch.cuda.synchronize(dev)
        torch.cuda.empty_cache()                # the warm-up steps' cached blocks go back: the graph owns its pool
        self.static_in = [d.clone() if torch.is_tensor(d) else d for d in data]
        self._src = [(d, d._version) if torch.is_tensor(d) else None for d in data]
        for o in self.optimizers:
            o.push_hyper()                      # outside the graph: stream-ordered in front of the launch
        streams.reset()                         # no eager-time stream may be pulled into the capture
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = self.fn(*self.static_in)
        streams.reset()                         # ... and no capture-time event leaks into later eager work
        if not isinstance(out, dict):
            raise TypeError('StepGraph: the step function must return a dict of outputs')
        self.static_out = out
        self.graph = g
        g.replay()                              # capture records, it does not execute: run this step now

    def run(self, *data):
        if not self.enabled or not graphs_enabled():
            return self.fn(*data)
        if self.graph is None:
            if self.calls < self.warmup:
                self.calls += 1
                return self.fn(*data)
            self._capture(data)                 # (the Python of the step ran once: host mirrors are already advanced)
            return self._outputs()
        self._load_inputs(data)
        for o in self.optimizers:
            o.push_hyper()
        self.graph.replay()
        for o in self.optimizers:
            o._hyper_pushed = False             # consumed by the replayed update kernel
        for h in self.replay_hooks:
            h()
        self.replays += 1
        return self._outputs()

    def reset(self):
        """Drop the captured graph (shapes / model structure changed): the next calls warm up and capture again."""
        self.graph = self.static_in = self.static_out = self._src = None
        self.calls = 0
