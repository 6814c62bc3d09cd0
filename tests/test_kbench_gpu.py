"""tools/kbench on the GPU: the convolution entry point of the C ABI checked BIT-EXACTLY without Python in the data
path (operands are multiples of 1/16, so fp32 accumulation is exact in any order and the expected output is the
bf16 rounding of an integer sum), the weight gradient compared == with host integer sums, and the BatchNorm finalize
launches against the host's fp64 arithmetic on the slab the device produced.  Reference semantics: paddle.nn.Conv2D /
paddle.nn.BatchNorm2D as used by resnetimagenet.py:114-153."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KBENCH = os.path.join(ROOT, 'tools', 'kbench')


def _kbench():
    """tools/kbench, rebuilt when it is missing or older than what it is compiled from / links against (a binary built
    against an earlier layout of the descriptors would hand the library garbage)."""
    deps = [os.path.join(ROOT, 'tools', 'kbench.cpp'), os.path.join(ROOT, 'include', 'passl_hip.h'),
            os.path.join(ROOT, 'passl_amd', 'lib', 'libpassl_hip.so')]
    stale = not os.path.exists(KBENCH) or any(os.path.exists(d) and os.path.getmtime(d) > os.path.getmtime(KBENCH)
                                              for d in deps)
    if stale:
        subprocess.run(['bash', os.path.join(ROOT, 'tools', 'build_kbench.sh')], check=True, capture_output=True,
                       text=True, timeout=900)
    return KBENCH


@pytest.mark.gpu
@pytest.mark.parametrize('options', [(), ('conv3x3_wave_rows=8',), ('conv3x3_wave=0',),
                                     ('igemm_persist=1', 'igemm_persist_grid=8')],
                         ids=['default', 'wave-8x8', 'no-wave', 'persistent-1x1'])
def test_conv_igemm_is_bit_exact_through_the_c_abi(options):
    """The default kernel selection (round 6: the wave-per-patch kernel on the 64 -> 64 3x3 cases), its 8 x 8-patch form,
    the selection without it, and the opt-in persistent form of the register-staged kernel with a grid of 8 workgroups
    (every workgroup walks many tiles): every output bit, every slab entry."""
    r = subprocess.run([_kbench(), 'check', *options], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert 'CHECK OK' in r.stdout
    if not options:
        assert 'kernel wave' in r.stdout
    if options == ('conv3x3_wave=0',):
        assert 'kernel wave' not in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize('options', [(), ('wgrad_halo=1',), ('wgrad_halo=0',)], ids=['default', 'halo1', 'per-tap'])
def test_conv_wgrad_is_exact_through_the_c_abi(options):
    """passl_hip_conv_wgrad compared == with host integer sums (1 / 7 / heuristic reduction slices; 3x3, 1x1, strided,
    ragged blocks): the default selection (the spatially tiled kernel on every 3x3 / stride-1 layer, incl. images its
    8 x 8 patches overhang), that kernel restricted to images whose sides are multiples of 8, and the per-tap kernels
    alone."""
    r = subprocess.run([_kbench(), 'wcheck', *options], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert 'WGRAD CHECK OK' in r.stdout


@pytest.mark.gpu
def test_bn_finalize_launches_equal_host_fp64_and_are_bit_reproducible():
    """passl_hip_bn_finalize / passl_hip_bn_bwd_finalize (one launch for any slab height since ABI 14) against the same
    arithmetic in fp64 on the host from the slab the device wrote: 128-row slabs as the conv epilogue writes them, a
    ragged last slab, more rows than one batch of loads, 3 rows; second run bit for bit."""
    r = subprocess.run([_kbench(), 'fincheck'], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert 'FINALIZE CHECK OK' in r.stdout


@pytest.mark.gpu
def test_stem_pool_reduce_second_form_agrees_with_the_first():
    """csrc/stem_pool.hip: the software-pipelined form of the fused BatchNorm + ReLU + max-pool backward's reduce pass
    against the first form on the same inputs (bf16 and fp32, odd image sides, 8 .. 256 channels, 5 .. 4096
    workgroups): the column sums of both slabs agree to fp32 rounding."""
    r = subprocess.run([_kbench(), 'poolcheck'], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert 'POOL CHECK OK' in r.stdout
