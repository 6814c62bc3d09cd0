"""ctypes binding of libpassl_hip.so (C ABI: include/passl_hip.h).

The library is the product's compute path.  There is NO fallback: if it cannot be loaded or a
tensor is not on a HIP device, calls raise.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), 'lib', 'libpassl_hip.so')

F32, BF16 = 0, 1
ABI_VERSION = 15          # include/passl_hip.h: PASSL_HIP_ABI_VERSION (the ctypes structs below mirror THAT layout)
_DT = {torch.float32: F32, torch.bfloat16: BF16}

c_p = C.c_void_p
c_i = C.c_int
c_l = C.c_int64
c_f = C.c_float


class ConvDesc(C.Structure):
    _fields_ = [('a', c_p), ('b', c_p), ('y', c_p), ('scale', c_p), ('shift', c_p),
                ('residual', c_p), ('stats', c_p),
                ('bnb_y', c_p), ('bnb_mask', c_p), ('bnb_mean', c_p), ('bnb_invstd', c_p),
                ('bnb_scale', c_p), ('bnb_shift', c_p), ('bnb_partial', c_p),
                ('N', C.c_int32), ('OP', C.c_int32), ('OQ', C.c_int32), ('NCOLS', C.c_int32),
                ('R', C.c_int32), ('S', C.c_int32), ('C', C.c_int32),
                ('IH', C.c_int32), ('IW', C.c_int32),
                ('sh', C.c_int32), ('sw', C.c_int32), ('ph', C.c_int32), ('pw', C.c_int32),
                ('a_sn', c_l), ('a_sh', c_l), ('a_sw', c_l),
                ('y_sn', c_l), ('y_sh', c_l), ('y_sw', c_l),
                ('relu', C.c_int32), ('dtype', C.c_int32), ('out_f32', C.c_int32),
                ('stats_tiles', C.c_int32), ('bnb_relu', C.c_int32), ('bnb_tile_off', C.c_int32),
                ('bnb2_y', c_p), ('bnb2_mean', c_p), ('bnb2_invstd', c_p), ('bnb2_partial', c_p)]


class WgradDesc(C.Structure):
    _fields_ = [('a', c_p), ('dy', c_p), ('dw', c_p),
                ('N', C.c_int32), ('OP', C.c_int32), ('OQ', C.c_int32), ('NCOLS', C.c_int32),
                ('R', C.c_int32), ('S', C.c_int32), ('C', C.c_int32),
                ('IH', C.c_int32), ('IW', C.c_int32),
                ('sh', C.c_int32), ('sw', C.c_int32), ('ph', C.c_int32), ('pw', C.c_int32),
                ('a_sn', c_l), ('a_sh', c_l), ('a_sw', c_l), ('dy_ld', c_l),
                ('ws', c_p), ('ws_floats', c_l),
                ('dtype', C.c_int32), ('splits', C.c_int32)]


class PackJob(C.Structure):
    _fields_ = [('src_off', c_l), ('dst_off', c_l),
                ('K', C.c_int32), ('R', C.c_int32), ('S', C.c_int32), ('C', C.c_int32),
                ('TR', C.c_int32), ('TS', C.c_int32),
                ('r_base', C.c_int32), ('r_step', C.c_int32),
                ('s_base', C.c_int32), ('s_step', C.c_int32),
                ('transpose', C.c_int32), ('c_pad', C.c_int32)]


# name -> (restype, argtypes); must list EVERY symbol declared in include/passl_hip.h
SIGNATURES = {
    'passl_hip_abi_version': (c_i, []),
    'passl_hip_strerror': (C.c_char_p, [c_i]),
    'passl_hip_set_option': (c_i, [C.c_char_p, c_i]),
    'passl_hip_last_igemm_kernel': (c_i, []),
    'passl_hip_ema_update': (c_i, [c_p, c_p, c_p, c_l, c_f, c_p]),
    'passl_hip_bn_fold': (c_i, [c_p, c_p, c_p, c_p, c_p, c_l, c_f, c_p, c_p, c_p]),
    'passl_hip_momentum_sgd': (c_i, [c_p, c_p, c_p, c_l, c_f, c_f, c_f, c_f, c_p]),
    'passl_hip_momentum_sgd_dev': (c_i, [c_p, c_p, c_p, c_l, c_p, c_f, c_f, c_f, c_p]),
    'passl_hip_cast_f32_to_bf16': (c_i, [c_p, c_p, c_l, c_p]),
    'passl_hip_pack_weights': (c_i, [c_p, c_p, c_i, c_p, c_p, c_p, c_i, c_p]),
    'passl_hip_nchw_to_nhwc_pad': (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    'passl_hip_conv_igemm': (c_i, [C.POINTER(ConvDesc), c_p]),
    'passl_hip_conv_wgrad': (c_i, [C.POINTER(WgradDesc), c_p]),
    'passl_hip_slab_reduce': (c_i, [c_p, c_p, c_l, c_i, c_i, c_p]),
    'passl_hip_bn_partial_floats': (c_l, [c_i, c_i, c_i]),
    'passl_hip_bn_stats': (c_i, [c_p, c_p, c_l, c_i, c_i, c_i, c_p]),
    'passl_hip_bn_finalize': (c_i, [c_p, c_i, c_l, c_i, c_i, c_p, c_p, c_p, c_p, c_f, c_f, c_p, c_p,
                                    c_p, c_p, c_p]),
    'passl_hip_bn_apply': (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_l, c_i, c_i, c_i, c_p]),
    'passl_hip_bn_bwd_reduce': (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_l, c_i, c_i, c_i,
                                      c_i, c_p]),
    'passl_hip_bn_bwd_finalize': (c_i, [c_p, c_i, c_l, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    'passl_hip_bn_moments': (c_i, [c_p, c_i, c_l, c_i, c_i, c_p, c_p]),
    'passl_hip_bn_finalize_moments': (c_i, [c_p, c_i, c_i, c_p, c_p, c_p, c_p, c_f, c_f, c_p, c_p, c_p, c_p, c_p]),
    'passl_hip_bn_bwd_sums': (c_i, [c_p, c_i, c_l, c_i, c_p, c_p]),
    'passl_hip_bn_bwd_finalize_sums': (c_i, [c_p, c_i, c_i, c_l, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    'passl_hip_bn_bwd_apply': (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_l, c_i, c_i, c_i,
                                     c_p]),
    'passl_hip_maxpool3x3s2_fwd': (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p]),
    'passl_hip_maxpool3x3s2_bwd': (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p]),
    'passl_hip_bn_relu_maxpool_blocks': (c_i, [c_i, c_i, c_i, c_i]),
    'passl_hip_bn_relu_maxpool_fwd': (c_i, [c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p]),
    'passl_hip_bn_relu_maxpool_bwd_reduce': (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i,
                                                   c_p]),
    'passl_hip_bn_relu_maxpool_bwd_apply': (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p]),
    'passl_hip_avgpool_fwd': (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_p]),
    'passl_hip_avgpool_bwd': (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_p]),
    'passl_hip_relu_bwd': (c_i, [c_p, c_p, c_p, c_l, c_i, c_p]),
    'passl_hip_colsum': (c_i, [c_p, c_p, c_l, c_i, c_i, c_p, c_l, c_p]),
    'passl_hip_colsum_acc': (c_i, [c_p, c_p, c_l, c_i, c_i, c_p, c_l, c_p]),
    'passl_hip_l2norm_fwd': (c_i, [c_p, c_p, c_p, c_i, c_i, c_f, c_p]),
    'passl_hip_l2norm_bwd': (c_i, [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_p]),
    'passl_hip_infonce_workspace_bytes': (c_l, [c_i, c_i]),
    'passl_hip_infonce_fwd': (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_f, c_p, c_p, c_p, c_p, c_p]),
    'passl_hip_infonce_bwd_workspace_bytes': (c_l, [c_i, c_i]),
    'passl_hip_infonce_bwd': (c_i, [c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_f, c_p, c_p, c_p]),
    'passl_hip_enqueue': (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_p]),
    'passl_hip_enqueue_dev': (c_i, [c_p, c_p, c_i, c_i, c_p, c_i, c_p]),
    'passl_hip_lars_momentum': (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_i, c_p, c_f, c_f,
                                      c_f, c_f, c_f, c_p]),
    'passl_hip_lars_momentum_dev': (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_i, c_p, c_p, c_f,
                                          c_f, c_f, c_f, c_p]),
    'passl_hip_larc_momentum_dev': (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_i, c_p, c_p, c_f,
                                          c_f, c_f, c_i, c_f, c_p]),
    'passl_hip_ntxent_fwd': (c_i, [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_f, c_p, c_p, c_p]),
    'passl_hip_ntxent_bwd': (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_f, c_p,
                                   c_p, c_p, c_p, c_p]),
    'passl_hip_layernorm_fwd': (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_l, c_i, c_f, c_i, c_p]),
    'passl_hip_layernorm_bwd': (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_l, c_i, c_i, c_p, c_l, c_p]),
    'passl_hip_layernorm_bwd_ws_floats': (c_l, [c_l, c_i]),
    'passl_hip_layernorm_param_reduce': (c_i, [c_p, c_l, c_i, c_p, c_p, c_p]),
    'passl_hip_gelu_fwd': (c_i, [c_p, c_p, c_l, c_i, c_p]),
    'passl_hip_gelu_bwd': (c_i, [c_p, c_p, c_p, c_l, c_i, c_p]),
    'passl_hip_tanh_fwd': (c_i, [c_p, c_p, c_l, c_i, c_p]),
    'passl_hip_tanh_bwd': (c_i, [c_p, c_p, c_p, c_l, c_i, c_p]),
    'passl_hip_attention_fwd': (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_i, c_i, c_p]),
    'passl_hip_attention_bwd': (c_i, [c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_i, c_i, c_p]),
    'passl_hip_quick_gelu_fwd': (c_i, [c_p, c_p, c_l, c_i, c_p]),
    'passl_hip_quick_gelu_bwd': (c_i, [c_p, c_p, c_p, c_l, c_i, c_p]),
    'passl_hip_embed_fwd': (c_i, [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p]),
    'passl_hip_embed_bwd': (c_i, [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_l, c_p, c_l, c_p]),
    'passl_hip_embed_bwd_acc_bytes': (c_l, [c_i, c_i]),
    'passl_hip_embed_bwd_ws_floats': (c_l, [c_i, c_i, c_i]),
    'passl_hip_gather_rows': (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_p]),
    'passl_hip_scatter_rows': (c_i, [c_p, c_p, c_p, c_i, c_l, c_i, c_i, c_p]),
    'passl_hip_eot_index': (c_i, [c_p, c_i, c_i, c_p, c_p]),
    'passl_hip_clip_logits_ws_floats': (c_l, [c_i, c_i]),
    'passl_hip_clip_logits_fwd': (c_i, [c_p, c_p, c_p, c_i, c_i, c_f, c_f, c_p, c_p, c_p]),
    'passl_hip_clip_logits_bwd': (c_i, [c_p, c_p, c_p, c_i, c_i, c_p, c_p, c_p, c_p, c_p]),
    'passl_hip_clip_scale': (c_i, [c_p, c_p, c_f, c_f, c_p]),
    'passl_hip_gemm_f32_nt': (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_p, c_p]),
    'passl_hip_gemm_f32_gx': (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_p]),
    'passl_hip_dot_acc': (c_i, [c_p, c_p, c_l, c_p, c_p, c_p]),
    'passl_hip_clip_ce_fwd': (c_i, [c_p, c_i, c_p, c_p, c_p, c_l, c_p]),
    'passl_hip_clip_ce_bwd': (c_i, [c_p, c_p, c_p, c_i, c_p, c_p]),
    'passl_hip_mae_mask': (c_i, [c_p, c_i, c_i, c_i, c_p, c_p, c_p, c_p]),
    'passl_hip_mae_gather': (c_i, [c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p]),
    'passl_hip_mae_gather_bwd': (c_i, [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p]),
    'passl_hip_mae_unshuffle': (c_i, [c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p]),
    'passl_hip_mae_unshuffle_bwd': (c_i, [c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_l, c_p]),
    'passl_hip_patchify': (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    'passl_hip_mae_loss_fwd': (c_i, [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_p, c_l, c_p]),
    'passl_hip_mae_loss_bwd': (c_i, [c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_p]),
    'passl_hip_adamw': (c_i, [c_p, c_p, c_p, c_p, c_l, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_p]),
    'passl_hip_adamw_dev': (c_i, [c_p, c_p, c_p, c_p, c_l, c_p, c_f, c_f, c_f, c_f, c_f, c_p]),
    'passl_hip_cosine_loss_fwd': (c_i, [c_p, c_p, c_i, c_i, c_f, c_p, c_p, c_p]),
    'passl_hip_cosine_loss_bwd': (c_i, [c_p, c_p, c_p, c_p, c_i, c_i, c_p, c_p]),
    'passl_hip_softmax_ce_fwd': (c_i, [c_p, c_p, c_i, c_i, c_p, c_p, c_p, c_l, c_p]),
    'passl_hip_softmax_ce_bwd': (c_i, [c_p, c_p, c_p, c_p, c_i, c_i, c_p, c_p]),
    'passl_hip_plan_create': (c_i, [C.POINTER(c_p)]),
    'passl_hip_plan_destroy': (c_i, [c_p]),
    'passl_hip_plan_record_begin': (c_i, [c_p]),
    'passl_hip_plan_cut': (c_i, [c_p]),
    'passl_hip_plan_record_end': (c_i, [c_p]),
    'passl_hip_plan_event_record': (c_i, [c_p, c_p]),
    'passl_hip_plan_stream_wait': (c_i, [c_p, c_p, c_i]),
    'passl_hip_plan_replay': (c_i, [c_p, c_i]),
    'passl_hip_plan_info': (c_l, [c_p, c_i]),
    'passl_hip_fill_zero': (c_i, [c_p, c_l, c_p]),
    'passl_hip_copy_bytes': (c_i, [c_p, c_p, c_l, c_p]),
    'passl_hip_cast_bf16_to_f32': (c_i, [c_p, c_p, c_l, c_p]),
    'passl_hip_unpad_add': (c_i, [c_p, c_p, c_l, c_i, c_i, c_i, c_i, c_i, c_p]),
    'passl_hip_prof_enable': (c_i, [c_i]),
    'passl_hip_prof_collect': (c_i, [c_i, C.POINTER(C.c_double), C.POINTER(c_l)]),
    'passl_hip_prof_collect_work': (c_i, [c_i, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    'passl_hip_prof_event_overhead': (c_i, [c_i, c_p, C.POINTER(C.c_double)]),
}

_lib = None
# status codes of include/passl_hip.h
OK, EINVAL, ELAUNCH, EUNSUPPORTED = 0, -1, -2, -3


class PasslHipError(RuntimeError):
    pass


def load(path=None):
    """dlopen the library and bind every symbol (raises if anything is missing)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise PasslHipError(
            'libpassl_hip.so not found at %s — build it with `python -m passl_amd.csrc.build` '
            '(the HIP path has no CPU fallback)' % p)
    lib = C.CDLL(p)
    lib.passl_hip_abi_version.restype = c_i
    if lib.passl_hip_abi_version() != ABI_VERSION:
        raise PasslHipError('%s is ABI %d, this binding is ABI %d: rebuild it (python -m passl_amd.csrc.build)'
                            % (p, lib.passl_hip_abi_version(), ABI_VERSION))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
        # PASSL_OPTIONS="name=value,name=value": passl_hip_set_option calls made once at load (kernel-selection A/Bs
        # without touching code; an unknown name is an error, not a silent no-op)
        for kv in filter(None, os.environ.get('PASSL_OPTIONS', '').split(',')):
            name, _, value = kv.partition('=')
            check(lib.passl_hip_set_option(name.strip().encode(), int(value)), 'PASSL_OPTIONS %s' % kv)
            from . import config
            config.mirror_library_option(name.strip(), int(value))      # (wgrad_halo: one switch, two readers)
        # the library reads PASSL_WGRAD_HALO itself; an explicit Python-side value (config.set_flag before load) wins
        from . import config
        if 'wgrad_halo' not in os.environ.get('PASSL_OPTIONS', ''):
            check(lib.passl_hip_set_option(b'wgrad_halo', int(config.wgrad_halo())), 'set_option wgrad_halo')
    return lib


def loaded():
    return _lib is not None


def set_option(name, value):
    """passl_hip_set_option + the Python-side twin of the option, when it has one."""
    check(load().passl_hip_set_option(name.encode(), int(value)), 'set_option %s' % name)
    from . import config
    config.mirror_library_option(name, int(value))


def check(rc, what=''):
    if rc != 0:
        msg = load().passl_hip_strerror(rc).decode()
        raise PasslHipError('%s failed: %s (status %d)' % (what or 'passl_hip call', msg, rc))


def dt(t):
    """passl_dtype code of a tensor / torch dtype."""
    d = t if isinstance(t, torch.dtype) else t.dtype
    try:
        return _DT[d]
    except KeyError:
        raise PasslHipError('unsupported dtype %s' % d)


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  Refuses host tensors: no CPU fallback."""
    if t is None:
        return None
    if not t.is_cuda:
        raise PasslHipError('passl_amd HIP op called with a %s tensor: the product path runs '
                            'on MI355X only (no CPU fallback)' % t.device)
    return t.data_ptr()


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def stream():
    """hipStream_t of torch's current stream on the current device.  torch.cuda.current_stream() builds a
    Stream object through several Python layers (9 us per call measured: ~4 ms of host time in a
    1500-launch CLIP step that is host-bound); the raw getter is the same handle in ~0.3 us."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream
