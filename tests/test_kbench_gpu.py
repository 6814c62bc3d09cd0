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
@pytest.mark.parametrize('options', [(), ('igemm_halo=1', 'igemm_halo_max_c=512'),
                                     ('igemm_halo=1', 'igemm_halo_max_c=512', 'igemm_halo_stages=4', 'igemm_halo_ck=64')],
                         ids=['default', 'halo', 'halo-4-stages-64'])
def test_conv_igemm_is_bit_exact_through_the_c_abi(options):
    r = subprocess.run([_kbench(), 'check', *options], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert 'CHECK OK' in r.stdout
    if options:
        # eight 3x3 / stride-1 cases x (ReLU, statistics, residual + ReLU, BatchNorm-backward statistics)
        assert r.stdout.count('kernel halo') == 32, r.stdout


@pytest.mark.gpu
def test_conv_wgrad_is_exact_through_the_c_abi_also_with_the_spatially_tiled_kernel():
    """passl_hip_conv_wgrad compared == with host integer sums (1 / 7 / heuristic reduction slices; 3x3, 1x1, strided,
    ragged blocks) for the product kernels and with the opt-in spatially tiled 3x3 kernel in its form for images whose
    sides are multiples of 8 (wgrad_halo=1: the instruction stream that was run on hardware in round 4)."""
    for options in ((), ('wgrad_halo=1',)):
        r = subprocess.run([_kbench(), 'wcheck', *options], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
        assert 'WGRAD CHECK OK' in r.stdout
