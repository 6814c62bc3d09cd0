"""The spatially tiled 3x3 kernel (passl_amd/csrc/conv_igemm_halo.hip, opt-in) without a GPU: its address
arithmetic lives in passl_amd/csrc/halo_geom.h, free of device intrinsics, and tests/emu/halo_emu.cpp executes the
kernel's data path lane by lane with those very functions — LDS-DMA pieces, ds_read_b128 fragment addresses, the
MFMA operand layout, the epilogue's row map — against a direct convolution on integer data, plus the bank-conflict
property of the halo layout for every start row.  Reference semantics: paddle.nn.Conv2D(3x3, stride 1, padding 1) as
used by resnetimagenet.py:121-124 (BottleneckBlock.conv2)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which('g++') is None, reason='needs g++')
def test_halo_kernel_addresses_reproduce_a_direct_convolution(tmp_path):
    exe = str(tmp_path / 'halo_emu')
    subprocess.run(['g++', '-O2', '-std=c++17', '-o', exe, os.path.join(ROOT, 'tests', 'emu', 'halo_emu.cpp')],
                   check=True, capture_output=True, text=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    assert 'EMULATION OK' in r.stdout
    # the layout claims of halo_geom.h
    assert 'pitch 144 B: 0 conflicts' in r.stdout and 'pitch 80 B: 0 conflicts' in r.stdout
    assert 'WRONG' not in r.stdout
