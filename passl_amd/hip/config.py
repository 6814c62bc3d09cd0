"""Process-wide settings of the HIP path: target device and compute dtype.

``set_device`` plays the role of ``paddle.set_device`` in the reference's Trainer
(passl_v110/engine/trainer.py:112-116): layers created afterwards allocate their parameters
there.  ``cpu`` is accepted so that host-side logic (registries, config, Trainer/hook plumbing,
gloo collectives) can be exercised without a GPU — the HIP layers themselves refuse to *run* on
a host tensor (passl_amd/hip/lib.py: no CPU fallback).
"""
import os

import torch

_state = {'device': None, 'dtype': torch.bfloat16,
          'fused_bn_stats': os.environ.get('PASSL_FUSED_BN_STATS', '1') != '0',
          'fuse_residual_grad': os.environ.get('PASSL_FUSE_RESIDUAL_GRAD', '1') != '0',
          'fused_bn_backward': os.environ.get('PASSL_FUSED_BN_BACKWARD', '1') != '0',
          'fused_bn_backward2': os.environ.get('PASSL_FUSED_BN_BACKWARD2', '1') != '0',
          'overlap': os.environ.get('PASSL_OVERLAP', '1') != '0',
          'fork_downsample': os.environ.get('PASSL_FORK_DOWNSAMPLE', '1') != '0',
          'side_reductions': os.environ.get('PASSL_SIDE_REDUCTIONS', '1') != '0',
          'fused_stem_pool': os.environ.get('PASSL_FUSED_STEM_POOL', '1') != '0',
          # pieces of the batch in which the fused stem pass hands its input gradient to the stem's weight gradient
          'stem_wgrad_parts': max(1, int(os.environ.get('PASSL_STEM_WGRAD_PARTS', '2') or 1)),
          'stem_tail_flush': os.environ.get('PASSL_STEM_TAIL_FLUSH', '1') != '0',
          'stem_wgrad_main_last': os.environ.get('PASSL_STEM_WGRAD_MAIN_LAST', '1') != '0',
          # a weight gradient over at least this many rows is handed to the side stream at once (0: always batched)
          'side_urgent_rows': int(os.environ.get('PASSL_SIDE_URGENT_ROWS', '500000') or 0),
          # the library reads the same variable (conv_wgrad_halo.inc): 0 off, 1 images with sides % 8 == 0, 2 all (default)
          'wgrad_halo': int(os.environ.get('PASSL_WGRAD_HALO', '2') or 0)}


def set_device(name):
    if isinstance(name, torch.device):
        _state['device'] = name
    elif name in ('gpu', 'cuda'):
        _state['device'] = torch.device('cuda', int(os.environ.get('PASSL_DEVICE_INDEX',
                                                                  os.environ.get('LOCAL_RANK', 0))))
        torch.cuda.set_device(_state['device'])
        _cap_memory(_state['device'])
    elif name == 'cpu':
        _state['device'] = torch.device('cpu')
    else:
        raise ValueError("device must be 'gpu' or 'cpu', got %r" % (name,))
    return _state['device']


_capped = set()


def _cap_memory(device):
    """Keep torch's caching allocator below PASSL_MEMORY_FRACTION (default 0.9 = 259 GB) of the device: the rest
    (29 GB) stays free for RCCL's communicator buffers and the driver on a data-parallel node.  The largest
    benchmarked configuration (SimCLR bs 512: 155 GB of live tensors) settles at 241-253 GB of cached blocks; with the
    cap a pool that wants more returns cached blocks and retries instead of taking the last free page.  Measured
    alternatives (profiles/r03_allocator_cap.txt): a 0.85 cap forces exactly that retry every step (273 -> 762
    ms/step), a garbage-collection threshold of 0.7 keeps 177 GB reserved but every collection is a
    device-synchronising free (351 ms/step), expandable segments change nothing on this build."""
    if not torch.cuda.is_available() or device.index in _capped:
        return
    frac = float(os.environ.get('PASSL_MEMORY_FRACTION', '0.9'))
    if 0.0 < frac < 1.0:
        torch.cuda.set_per_process_memory_fraction(frac, device)
    _capped.add(device.index)


def get_device():
    if _state['device'] is None:
        set_device('gpu' if torch.cuda.is_available() else 'cpu')
    return _state['device']


def set_compute_dtype(dt):
    if isinstance(dt, str):
        dt = {'bf16': torch.bfloat16, 'bfloat16': torch.bfloat16, 'fp32': torch.float32,
              'float32': torch.float32}[dt]
    assert dt in (torch.bfloat16, torch.float32)
    _state['dtype'] = dt


def get_compute_dtype():
    return _state['dtype']


def fused_bn_stats():
    """BatchNorm statistics accumulated by the producing conv's epilogue (bf16 only) instead of a
    separate pass over the conv output."""
    return _state['fused_bn_stats']


def fuse_residual_grad():
    """Residual-fork gradient added in conv1's data-gradient epilogue (nn.GradSlot) instead of an
    autograd add kernel."""
    return _state['fuse_residual_grad']


def fused_bn_backward():
    """BatchNorm backward statistics (sum g, sum g*xhat) written by the epilogue of the data-gradient
    launch that produces the BatchNorm output's gradient (bf16 only) instead of a separate pass."""
    return _state['fused_bn_backward']


def fused_bn_backward2():
    """With `fused_bn_backward`: the data-gradient launch behind a downsample block also reduces the backward
    statistics of the downsample branch's BatchNorm (same gradient, second (y, mean, invstd): conv desc bnb2_*)."""
    return _state['fused_bn_backward2']


def overlap():
    """Independent work on a second HIP stream: weight-gradient launches next to the data-gradient /
    BatchNorm-backward chain (hip/streams.py)."""
    return _state['overlap']


def fork_downsample():
    """With `overlap`: the downsample branch of a bottleneck block (1x1 conv + BatchNorm of the block
    input) runs on the second HIP stream next to conv1..conv3, forward and backward."""
    return _state['fork_downsample']


def side_reductions():
    """With `overlap`: the small parameter-gradient reductions of a backward node (a Linear's bias column sums, the
    fold of LayerNorm's d-gamma / d-beta partials) run on the second HIP stream, off the data-gradient chain."""
    return _state['side_reductions']


def fused_stem_pool():
    """Training-mode BatchNorm + ReLU + max-pool of the ResNet stem as one pass per direction (csrc/stem_pool.hip):
    the BatchNorm output and the pool's input gradient are never written."""
    return _state['fused_stem_pool']


def stem_wgrad_parts():
    """With `fused_stem_pool` and `overlap`: the stem's BatchNorm-backward apply pass and the stem's weight gradient —
    the last two launches of a backward pass, the second reading what the first writes — run over this many pieces
    of the batch, the weight gradient of a piece on the side stream next to the apply pass of the next (1 = whole)."""
    return _state['stem_wgrad_parts']


def stem_wgrad_main_last():
    """With `stem_wgrad_parts` > 1: the LAST piece's weight gradient runs on the main stream (which has nothing else left
    in the backward pass) while the side stream still works on the piece before it."""
    return _state['stem_wgrad_main_last']


def side_urgent_rows():
    """Weight gradients over at least this many rows (N * OH * OW: the first trunk stage at batch 256) skip the batched
    hand-off to the side stream (hip/streams.py: PASSL_SIDE_BATCH): they are the longest launches of the side stream and
    the last ones of a backward pass — queued in fours they all ran behind the main chain's end
    (profiles/r05_trace_timeline_tail.txt)."""
    return _state['side_urgent_rows']


def stem_tail_flush():
    """The fused stem backward first hands the weight gradients still queued for the side stream over (they then run
    next to its two passes instead of behind them, in front of the optimizer)."""
    return _state['stem_tail_flush']


def wgrad_halo():
    """The spatially tiled 3x3 weight-gradient kernel takes eligible launches (2, the default: every 3x3 / stride-1
    layer; 1: images whose sides are multiples of 8; 0: off); the slice count of those launches is then chosen for ITS
    grid (one workgroup per 64 x 64 block of dW and slice, all nine taps).  Measured (profiles/r05_kbench_first_call.txt):
    64->64 @56 78 vs 147 us, 128->128 @28 87 vs 94, 256->256 @14 84 vs 93, 512->512 @7 84 vs 104."""
    return _state['wgrad_halo']


_BOOL_FLAGS = ('fused_bn_stats', 'fuse_residual_grad', 'fused_bn_backward', 'overlap', 'fork_downsample',
               'side_reductions', 'fused_stem_pool', 'fused_bn_backward2', 'stem_tail_flush', 'stem_wgrad_main_last')
_INT_FLAGS = ('stem_wgrad_parts', 'side_urgent_rows', 'wgrad_halo')


def set_flag(name, value):
    """Switch a scheduling / fusion option at run time (tests, A/B runs).  ``wgrad_halo`` is ONE switch shared with the
    library (its kernel choice and the slice count chosen here must agree: round-5 advisor finding): setting it here
    also sets it there."""
    if name in _BOOL_FLAGS:
        _state[name] = bool(value)
    elif name in _INT_FLAGS:
        _state[name] = max(1, int(value)) if name == 'stem_wgrad_parts' else int(value)
        if name == 'wgrad_halo':
            from . import lib as L
            if L.loaded():
                L.check(L.load().passl_hip_set_option(b'wgrad_halo', int(value)), 'set_option wgrad_halo')
    else:
        raise KeyError('unknown flag %r (known: %s)' % (name, ', '.join(_BOOL_FLAGS + _INT_FLAGS)))


def mirror_library_option(name, value):
    """hip/lib.py tells us about a library option it has just set (PASSL_OPTIONS, set_option) that has a Python-side twin."""
    if name == 'wgrad_halo':
        _state['wgrad_halo'] = int(value)
