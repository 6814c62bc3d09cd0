"""The C-ABI library: loads without a GPU, exports every symbol include/passl_hip.h declares,
the ctypes structs have the C layout, and argument errors are reported before any launch."""
import ctypes
import os
import re
import subprocess

import pytest

from passl_amd.hip import lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'passl_hip.h')


@pytest.fixture(scope='module')
def lib():
    if not os.path.exists(L.LIB_PATH):
        from passl_amd.csrc.build import build
        build()
    return L.load()


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(passl_hip_\w+)\s*\(', src)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    syms = declared_symbols()
    assert len(syms) >= 29
    for s in syms:
        assert hasattr(lib, s), 'not exported: ' + s
        assert s in L.SIGNATURES, 'declared in the header but not bound in lib.py: ' + s
    assert sorted(L.SIGNATURES) == syms, 'lib.py binds symbols the header does not declare'


def test_abi_version_and_strerror(lib):
    assert lib.passl_hip_abi_version() == L.ABI_VERSION == 15
    assert ('#define PASSL_HIP_ABI_VERSION %d' % L.ABI_VERSION) in open(HEADER).read()
    assert b'invalid' in lib.passl_hip_strerror(-1)
    assert lib.passl_hip_strerror(0) == b'ok'


def test_struct_layouts_match_the_header(tmp_path):
    src = tmp_path / 'sz.c'
    src.write_text('#include "%s"\n#include <stdio.h>\n#include <stddef.h>\nint main(){'
                   'printf("%%zu %%zu %%zu %%zu %%zu %%zu\\n", sizeof(passl_pack_job), sizeof(passl_conv_desc),'
                   'sizeof(passl_wgrad_desc), offsetof(passl_conv_desc, a_sn), offsetof(passl_conv_desc, relu),'
                   'offsetof(passl_wgrad_desc, dy_ld)); return 0; }\n' % HEADER)
    exe = tmp_path / 'sz'
    subprocess.check_call(['gcc', str(src), '-o', str(exe)])
    out = subprocess.check_output([str(exe)]).decode().split()
    assert [int(v) for v in out] == [ctypes.sizeof(L.PackJob), ctypes.sizeof(L.ConvDesc),
                                     ctypes.sizeof(L.WgradDesc), L.ConvDesc.a_sn.offset,
                                     L.ConvDesc.relu.offset, L.WgradDesc.dy_ld.offset]


def test_argument_errors_are_reported_without_a_gpu(lib):
    # null pointers / bad shapes are rejected before anything is launched
    assert lib.passl_hip_ema_update(None, None, None, 16, 0.999, None) == -1
    assert lib.passl_hip_momentum_sgd(None, None, None, 16, 0.1, 0.9, 0.0, 1.0, None) == -1
    assert lib.passl_hip_infonce_fwd(None, None, None, 8, 128, 1024, 0.2, None, None, None, None, None) == -1
    d = L.ConvDesc()
    assert lib.passl_hip_conv_igemm(ctypes.byref(d), None) == -1
    w = L.WgradDesc()
    assert lib.passl_hip_conv_wgrad(ctypes.byref(w), None) == -1
    assert lib.passl_hip_infonce_workspace_bytes(256, 65536) == (512 + 1) * 256 * 16
    assert lib.passl_hip_infonce_bwd_workspace_bytes(256, 65536) == (512 + 1) * 256 * 128 * 4


def test_host_tensors_are_refused():
    import torch
    from passl_amd.hip import ops
    with pytest.raises(L.PasslHipError):
        ops.ema_update(torch.zeros(16), torch.zeros(16), 0.9)


def test_every_entry_point_rejects_null_arguments(lib):
    """Every compute entry point validates its arguments before touching the device: all-NULL / zero
    arguments yield an error status (never a crash, never a launch) — checked without a GPU."""
    import ctypes as C
    skip = {'passl_hip_abi_version', 'passl_hip_strerror', 'passl_hip_set_option', 'passl_hip_prof_enable',
            'passl_hip_last_igemm_kernel',
            'passl_hip_infonce_workspace_bytes', 'passl_hip_infonce_bwd_workspace_bytes',
            'passl_hip_bn_partial_floats', 'passl_hip_bn_relu_maxpool_blocks',
            'passl_hip_clip_logits_ws_floats', 'passl_hip_layernorm_bwd_ws_floats',
            'passl_hip_embed_bwd_acc_bytes', 'passl_hip_embed_bwd_ws_floats'}
    checked = 0
    for name, (res, args) in sorted(L.SIGNATURES.items()):
        if name in skip:
            continue
        zeros = []
        for a in args:
            if a is L.c_f:
                zeros.append(0.0)
            elif a in (L.c_i, L.c_l):
                zeros.append(0)
            else:
                zeros.append(None)
        rc = getattr(lib, name)(*zeros)
        assert rc in (-1, -3), '%s(all null) returned %r' % (name, rc)
        checked += 1
    assert checked >= 50
    # shapes outside a kernel's envelope are PASSL_EUNSUPPORTED, not a silent wrong answer
    buf = (C.c_float * 64)()
    p = C.cast(buf, C.c_void_p)
    assert lib.passl_hip_attention_fwd(p, p, p, 1, 300, 1, 64, 0.125, 0, L.F32, None) == -3     # T > 208
    assert lib.passl_hip_attention_fwd(p, p, p, 1, 16, 1, 48, 0.125, 0, L.F32, None) == -3      # head dim
    assert lib.passl_hip_layernorm_bwd(p, p, p, p, p, None, p, p, p, 4, 4096, L.F32, p, 1 << 20, None) == -1  # C > 2048
    # reductions have no atomic fall-back: the workspace is mandatory, and its size is published
    assert lib.passl_hip_layernorm_bwd(p, p, p, p, p, None, p, p, p, 64, 16, L.F32, None, 0, None) == -1
    assert lib.passl_hip_layernorm_bwd_ws_floats(50432, 768) == 1030 * 2 * 768
    assert lib.passl_hip_embed_bwd_acc_bytes(49408, 512) == 49408 * 512 * 8 + 49408 * 4 + 16
    assert lib.passl_hip_embed_bwd_ws_floats(256, 77, 512) == 2 * 77 * 512
    assert lib.passl_hip_set_option(b'no_such_option', 1) == -1


def test_tuning_options_validate_their_values(lib):
    """passl_hip_set_option: known names accept their documented values and reject others; unknown
    names are refused (no launch involved: runs without a GPU)."""
    ok = [(b'bn_stream_unroll', 0), (b'bn_stream_unroll', 2), (b'bn_stream_unroll', 8), (b'bn_stream_unroll', 4),
          (b'stem_kernel', 0), (b'stem_kernel', 1), (b'igemm_ring_bm', 256), (b'igemm_ring_bm', 128),
          (b'igemm_ring_min_nk', 8), (b'igemm_8p_dense', 1), (b'igemm_8p_dense', 2), (b'igemm_8p_dense', 0)]
    for name, v in ok:
        assert lib.passl_hip_set_option(name, v) == 0, (name, v)
    bad = [(b'bn_stream_unroll', 3), (b'igemm_ring_bm', 7), (b'igemm_8p_dense', 3), (b'no_such_option', 1)]
    for name, v in bad:
        assert lib.passl_hip_set_option(name, v) != 0, (name, v)


def test_step_plan_api_without_launches(lib):
    """The native step plan's bookkeeping (include/passl_hip.h "native step plans") without a GPU: a plan records
    once, one plan at a time, cut() numbers the segments, replaying an empty segment is a no-op, bad handles /
    segments / event ids are refused."""
    import ctypes as C
    h = C.c_void_p()
    assert lib.passl_hip_plan_create(C.byref(h)) == 0 and h.value
    assert lib.passl_hip_plan_replay(h, 0) == -1                      # nothing recorded yet
    assert lib.passl_hip_plan_cut(h) == -1                            # not recording
    assert lib.passl_hip_plan_record_begin(h) == 0
    h2 = C.c_void_p()
    assert lib.passl_hip_plan_create(C.byref(h2)) == 0
    assert lib.passl_hip_plan_record_begin(h2) == -1                  # one recording per process at a time
    assert lib.passl_hip_plan_stream_wait(h, None, 0) == -1           # no such event
    assert lib.passl_hip_plan_cut(h) == 1
    assert lib.passl_hip_plan_cut(h) == 2
    assert lib.passl_hip_plan_record_end(h) == 0
    assert lib.passl_hip_plan_record_begin(h) == -1                   # a plan records once
    assert lib.passl_hip_plan_info(h, 0) == 3 and lib.passl_hip_plan_info(h, 1) == 0
    for seg in range(3):
        assert lib.passl_hip_plan_replay(h, seg) == 0
    assert lib.passl_hip_plan_replay(h, 3) == -1
    assert lib.passl_hip_plan_info(h, 6) == 1
    assert lib.passl_hip_plan_record_begin(h2) == 0 and lib.passl_hip_plan_record_end(h2) == 0
    assert lib.passl_hip_plan_destroy(h) == 0 and lib.passl_hip_plan_destroy(h2) == 0
