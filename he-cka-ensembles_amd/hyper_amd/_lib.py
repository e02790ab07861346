"""ctypes binding of libhecka.so (the C-ABI declared in include/hecka.h).

There is deliberately NO fallback: if the shared library is missing or a call fails, the
product path raises.  Build it with `make -C he-cka-ensembles_amd/csrc` (or `__graft_entry__.build()`).
"""
import ctypes
import os

_PATH = os.environ.get('HECKA_LIB_PATH') or os.path.join(os.path.dirname(os.path.abspath(__file__)), 'lib', 'libhecka.so')
_lib = None

_vp, _i, _ll, _sz, _d = ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong, ctypes.c_size_t, ctypes.c_double

# name -> (restype, argtypes); must list every symbol include/hecka.h declares
SIGNATURES = {
    'hecka_version': (_i, []),
    'hecka_last_error': (ctypes.c_char_p, []),
    'hecka_colmean_f32': (_i, [_vp, _i, _i, _i, _vp, _vp, _vp]),
    'hecka_colmean_multi_f32': (_i, [_i, _vp, _i, _i, _vp, _vp, _vp, _vp]),
    'hecka_gram_center_f32': (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _i, _vp]),
    'hecka_pairwise_workspace_bytes': (_sz, [_i, _i, _ll]),
    'hecka_pairwise_f32': (_i, [_vp, _i, _vp, _i, _ll, _vp, _vp, _sz, _vp]),
    'hecka_noalloc_init': (_i, []),
    'hecka_pairwise_noalloc_f32': (_i, [_vp, _i, _vp, _i, _ll, _vp, _vp]),
    'hecka_head_fwd_f32': (_i, [_vp, _vp, _i, _i, _i, _i, _d, _d, _d, _i, _vp, _vp, _d, _vp]),
    'hecka_head_bwd_f32': (_i, [_vp, _vp, _i, _i, _i, _i, _d, _d, _d, _i, _vp, _vp, _d, _vp, _vp]),
    'hecka_coef_gamma_f32': (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    'hecka_combine_f32': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    'hecka_rowmix_f32': (_i, [_vp, _vp, _i, _i, _ll, _vp, _vp]),
    'hecka_center_rows_f32': (_i, [_vp, _ll, _i, _vp, _vp, _vp]),
    'hecka_symmetrize_f32': (_i, [_vp, _i, _i, _i, _vp, _vp]),
    'hecka_rbf_bandwidth_f32': (_i, [_vp, _i, _i, ctypes.c_float, _vp, _vp]),
    'hecka_gram_transform_f32': (_i, [_vp, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp]),
    'hecka_gram_transform_bwd_f32': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    'hecka_bgemm_f32': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _ll, _ll, _i, ctypes.c_float, _vp, ctypes.c_float, _vp]),
    'hecka_bgemm_splitk_workspace_bytes': (ctypes.c_size_t, [_i, _i, _i, _i]),
    'hecka_bgemm_splitk_f32': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _ll, _ll, _i, ctypes.c_float, _vp, ctypes.c_size_t, _vp]),
    'hecka_im2col_f32': (_i, [_vp] + [_i] * 14 + [_vp, _vp]),
    'hecka_col2im_f32': (_i, [_vp] + [_i] * 14 + [_vp, _vp]),
    'hecka_gram_center_multi_f32': (_i, [_i, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _i, _vp]),
    'hecka_wbound_f32': (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    'hecka_colmean_x16': (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    'hecka_gram_center_multi_x16': (_i, [_i, _vp, _i, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _vp]),
    'hecka_gram_center_multi_workspace_bytes': (_sz, [_i, _i, _i, _vp]),
    'hecka_gram_center_spec_eligible': (_i, [_i, _i, _i, _vp]),
    'hecka_gram_center_spec_workspace_bytes': (_sz, [_i, _i, _i, _vp]),
    'hecka_gram_center_spec_multi_f32': (_i, [_i, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    'hecka_gram_bwd_x16': (_i, [_vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _vp, _i, _vp]),
    'hecka_head_fwd_batched_f32': (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'hecka_head_bwd_batched_f32': (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'hecka_coef_gamma_batched_f32': (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    'hecka_wbound_batched_f32': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    'hecka_pairwise_self_batched_f32': (_i, [_vp, _i, _i, _ll, _vp, _vp, ctypes.c_size_t, _vp]),
    'hecka_absmax_rows_f32': (_i, [_vp, _i, _ll, _vp, _vp]),
    'hecka_bgemm_x3_eligible': (_i, [_i, _i, _i, _i]),
    'hecka_bgemm_x3_workspace_bytes': (_sz, [_i, _i, _i, _i]),
    'hecka_bgemm_x3_f32': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _ll, _ll, _i, ctypes.c_float, _vp, _vp, _i, _vp, _sz, _vp]),
    'hecka_batchnorm_act_fwd_f32': (_i, [_vp, _vp, _i, _i, _i, _ll, _vp, _vp, _vp, _vp, _i, ctypes.c_float, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    'hecka_batchnorm_act_fwd_packed_f32': (_i, [_vp, _vp, _i, _i, _i, _ll, _vp, _vp, _i, _vp, _vp, _i, ctypes.c_float, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_float, _vp, _vp]),
    'hecka_batchnorm_act_bwd_packed_f32': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _ll, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'hecka_conv_direct_eligible': (_i, [_i, _i, _i, _i]),
    'hecka_conv_direct_f32': (_i, [_vp, _ll, _vp, _vp, ctypes.c_float, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    'hecka_conv_direct_wgrad_f32': (_i, [_vp, _ll, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    'hecka_gemv_t_workspace_bytes': (_sz, [_i, _i]),
    'hecka_gemv_t_f32': (_i, [_vp, _vp, _i, _i, _vp, _vp, _sz, _vp]),
    'hecka_gemv_f32': (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    'hecka_ger_f32': (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp]),
    'hecka_groupnorm_fwd_f32': (_i, [_vp, _i, _i, _i, _vp, _vp, ctypes.c_float, _i, _vp, _vp, _vp, _vp]),
    'hecka_groupnorm_bwd_f32': (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    'hecka_crater_f32': (_i, [_vp, _vp, _vp, _ll, _vp]),
    'hecka_crater_maxpool2_f32': (_i, [_vp, _vp, _vp, _ll, _i, _i, _vp]),
    'hecka_batchnorm_act_bwd_f32': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _ll, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    'hecka_conv_x3_eligible': (_i, [_i, _i, _i, _i, _i, _i, _i]),
    'hecka_conv_x3_strided_eligible': (_i, [_i, _i, _i, _i, _i, _i, _i, _i, _i]),
    'hecka_conv_x3_strided_f32': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _ll, _ll, _i, _i, _i, _i, _i, _i, _i, _i, ctypes.c_float, _vp, _vp, _i, _vp]),
    'hecka_conv_x3_placed_f32': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _ll, _ll, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, ctypes.c_float, _vp, _vp, _i, _vp]),
    'hecka_conv_x3_tapmajor_f32': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _ll, _ll, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, ctypes.c_float, _vp, _vp, _i, _vp]),
    'hecka_conv_weights_tapmajor_f32': (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp]),
    'hecka_set_layer_products': (_i, [_i]),
    'hecka_get_layer_products': (_i, []),
    'hecka_conv_wgrad_x3_strided_eligible': (_i, [_i, _i, _i, _i, _i, _i, _i, _i, _i]),
    'hecka_conv_wgrad_x3_strided_workspace_bytes': (_sz, [_i, _i, _i, _i, _i, _i, _i, _i, _i]),
    'hecka_conv_wgrad_x3_strided_f32': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _ll, _i, _i, _i, _i, _i, _i, _i, _i, ctypes.c_float, _vp, _vp, _i, _vp, _sz, _vp]),
    'hecka_conv_x3_f32': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _ll, _ll, _i, _i, _i, _i, ctypes.c_float, _vp, _vp, _i, _vp]),
    'hecka_conv_wgrad_x3_eligible': (_i, [_i, _i, _i, _i, _i, _i, _i]),
    'hecka_conv_wgrad_x3_workspace_bytes': (_sz, [_i, _i, _i, _i, _i, _i, _i]),
    'hecka_conv_wgrad_x3_f32': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _ll, _i, _i, _i, _i, ctypes.c_float, _vp, _vp, _i, _vp, _sz, _vp]),
    'hecka_combine_slice_f32': (_i, [_vp, _vp, _vp, _i, _i, _i, _ll, _ll, _i, _vp, _vp]),
    'hecka_combine_slice_batched_f32': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _ll, _ll, _i, _vp, _vp]),
    'hecka_batchnorm_fwd_f32': (_i, [_vp, _i, _i, _i, _ll, _vp, _vp, _vp, _vp, _i, ctypes.c_float, _vp, _vp, _vp, _vp, _vp]),
    'hecka_batchnorm_bwd_f32': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _ll, _i, _vp, _vp, _vp, _vp]),
    'hecka_gram_bwd_f32': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _i, _vp]),
    'hecka_gram_bwd_multi_eligible': (_i, [_i, _i, _i, _vp, _i]),
    'hecka_gram_bwd_multi_f32': (_i, [_i, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _i, _i, _i, _vp]),
    'hecka_config_reload': (None, []),
    'hecka_blocked_rows_eligible': (_i, [_i, _ll, _i]),
    'hecka_pairwise_self_blocked_f32': (_i, [_vp, _i, _i, _ll, _i, _vp, _vp, _sz, _vp]),
    'hecka_combine_slice_blocked_f32': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _ll, _ll, _i, _i, _vp, _vp]),
}


def library_path():
    return _PATH


def lib():
    """The loaded library; raises RuntimeError (never falls back) when it cannot be loaded."""
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            raise RuntimeError('libhecka.so not found at %s - build it with `make -C he-cka-ensembles_amd/csrc`; '
                               'hyper_amd has no non-HIP fallback' % _PATH)
        # torch bundles its own libamdhip64; load it first so libhecka binds to the SAME HIP runtime
        # (two runtimes in one process do not share devices or allocations)
        import torch  # noqa: F401
        handle = ctypes.CDLL(_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


def reload_config():
    """Re-read the library's HECKA_* environment knobs (they are read once, at load time)."""
    lib().hecka_config_reload()


def check(rc, what):
    if rc != 0:
        msg = lib().hecka_last_error()
        raise RuntimeError('libhecka %s failed (%d): %s' % (what, rc, msg.decode() if msg else '?'))
