"""The spatially tiled 3x3 weight-gradient kernel (passl_amd/csrc/conv_wgrad_halo.inc) without a GPU: its address
arithmetic lives in passl_amd/csrc/wgrad_halo_geom.h, free of device intrinsics, and tests/emu/wgrad_halo_emu.cpp
executes the kernel's data path lane by lane with those very functions against a direct weight gradient on integer
data.  Reference semantics: autograd of paddle.nn.Conv2D(3x3, stride 1, padding 1) as used by
resnetimagenet.py:121-124 (BottleneckBlock.conv2).  (The forward kernel of the same idea was measured level with /
slower than the ring kernel in rounds 4-5 and removed: profiles/r05_negative_results.txt.)"""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which('g++') is None, reason='needs g++')
def test_wgrad_halo_kernel_addresses_reproduce_a_direct_weight_gradient(tmp_path):
    """The spatially tiled 3x3 weight-gradient kernel (passl_amd/csrc/conv_wgrad_halo.inc): LDS-DMA pieces in the
    kernel's lane-static + per-patch form, ds_read_b64_tr_b16 with its 16-lane transposition, MFMA operand layout and the
    accumulator -> dW map (passl_amd/csrc/wgrad_halo_geom.h), incl. images that the 8 x 8 patches overhang and several
    reduction slices; autograd of paddle.nn.Conv2D w.r.t. its weight (resnetimagenet.py:121-124)."""
    exe = str(tmp_path / 'wgrad_halo_emu')
    subprocess.run(['g++', '-O2', '-std=c++17', '-o', exe, os.path.join(ROOT, 'tests', 'emu', 'wgrad_halo_emu.cpp')],
                   check=True, capture_output=True, text=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    assert 'EMULATION OK' in r.stdout and 'WRONG' not in r.stdout and 'differs' not in r.stdout
    assert 'under the transposing read: 0 conflicts' in r.stdout


def test_wgrad_slice_choice_follows_the_kernel_that_takes_the_launch(monkeypatch):
    """Host logic (passl_amd/hip/ops.py: wgrad_slices): the spatially tiled kernel's own grid shape for the launches
    it takes (every 3x3 / stride-1 bf16 layer by default), the per-tap kernels' heuristic for everything else."""
    import torch
    from passl_amd.hip import config, ops, plan as P
    g3 = P.ConvGeom(cin=64, cout=64, k=3, stride=1, pad=1)
    d56, d28 = P.fwd_desc(g3, 256, 56, 56), P.fwd_desc(g3, 256, 28, 28)
    d1 = P.fwd_desc(P.ConvGeom(cin=64, cout=256, k=1, stride=1, pad=0), 256, 56, 56)
    monkeypatch.setitem(config._state, 'wgrad_halo', 0)
    base = {id(d): ops.wgrad_slices(d, torch.bfloat16) for d in (d56, d28, d1)}
    assert base[id(d56)] == 51                                   # 5 tiles of 128 x 128 -> 256 // 5 (round 6: one
    #                                                              workgroup per CU for convolutional layers)
    monkeypatch.setitem(config._state, 'wgrad_halo', 1)
    assert ops.wgrad_slices(d56, torch.bfloat16) == 256          # one 64 x 64 block: 256 slices (best INSIDE the step)
    assert ops.wgrad_slices(d28, torch.bfloat16) == base[id(d28)]   # 28 is not a multiple of 8: not taken in mode 1
    assert ops.wgrad_slices(d1, torch.bfloat16) == base[id(d1)]
    assert ops.wgrad_slices(d56, torch.float32) != 256
    monkeypatch.setitem(config._state, 'wgrad_halo', 2)
    assert ops.wgrad_slices(d28, torch.bfloat16) == 256
