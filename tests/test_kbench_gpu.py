"""tools/kbench on the GPU: the convolution entry point of the C ABI checked BIT-EXACTLY without Python in the data
path (operands are multiples of 1/16, so fp32 accumulation is exact in any order and the expected output is the
bf16 rounding of an integer sum) — for the default kernel selection and for the opt-in spatially tiled 3x3 kernel
(conv_igemm_halo.hip).  Reference semantics: paddle.nn.Conv2D as used by resnetimagenet.py:114-131."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KBENCH = os.path.join(ROOT, 'tools', 'kbench')


def _kbench():
    if not os.path.exists(KBENCH):
        subprocess.run(['bash', os.path.join(ROOT, 'tools', 'build_kbench.sh')], check=True, capture_output=True,
                       text=True, timeout=900)
    return KBENCH


@pytest.mark.gpu
@pytest.mark.parametrize('options', [(), ('igemm_halo=1', 'igemm_halo_max_c=512')], ids=['default', 'halo'])
def test_conv_igemm_is_bit_exact_through_the_c_abi(options):
    r = subprocess.run([_kbench(), 'check', *options], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert 'CHECK OK' in r.stdout
    if options:
        assert 'kernel halo' in r.stdout, r.stdout          # the 3x3 / stride-1 cases went through the opt-in kernel


@pytest.mark.gpu
def test_conv_wgrad_is_exact_through_the_c_abi_also_with_the_spatially_tiled_kernel():
    """passl_hip_conv_wgrad compared == with host integer sums (1 / 7 / heuristic reduction slices; 3x3, 1x1, strided,
    ragged blocks): the product kernels on the images whose sides are not multiples of 8, the opt-in spatially tiled 3x3
    kernel on the others (wgrad_halo=1: the instruction stream and the command line that ran on hardware in round 4)."""
    r = subprocess.run([_kbench(), 'wcheck', 'wgrad_halo=1'], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert 'WGRAD CHECK OK' in r.stdout
