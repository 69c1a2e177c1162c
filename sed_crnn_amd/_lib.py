"""ctypes binding of libsedcrnn.so (the C ABI declared in include/sedcrnn.h).

There is NO fallback: if the shared library is missing or a call fails this module raises.
The product path never routes through torch.nn compute ops or the CPU oracle.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# SED_CRNN_LIB: another build of the same library (A/B measurements of a kernel change on one box); default in-tree
LIB_PATH = os.environ.get("SED_CRNN_LIB") or os.path.join(HERE, "libsedcrnn.so")

SED_MAX_CONV = 4
SED_MAX_GRU = 4
SED_MAX_DENSE = 4

_fp = C.c_void_p          # device pointers travel as integers
_stream = C.c_void_p


class NetCfg(C.Structure):
    _fields_ = [
        ("B", C.c_int), ("Cin", C.c_int), ("F", C.c_int), ("T", C.c_int),
        ("n_conv", C.c_int),
        ("C", C.c_int * SED_MAX_CONV),
        ("pool_f", C.c_int * SED_MAX_CONV), ("pool_t", C.c_int * SED_MAX_CONV),
        ("drop_p", C.c_float * SED_MAX_CONV),
        ("n_gru", C.c_int), ("H", C.c_int * SED_MAX_GRU),
        ("n_dense", C.c_int), ("D", C.c_int * SED_MAX_DENSE),
        ("bn_eps", C.c_float), ("bn_momentum", C.c_float),
        ("conv_mode", C.c_int),
        ("flags", C.c_int),
    ]


class NetParams(C.Structure):
    _fields_ = [
        ("conv_w", _fp * SED_MAX_CONV), ("conv_b", _fp * SED_MAX_CONV),
        ("bn_g", _fp * SED_MAX_CONV), ("bn_b", _fp * SED_MAX_CONV),
        ("bn_rm", _fp * SED_MAX_CONV), ("bn_rv", _fp * SED_MAX_CONV),
        ("gru_wih", (_fp * 2) * SED_MAX_GRU), ("gru_whh", (_fp * 2) * SED_MAX_GRU),
        ("gru_bih", (_fp * 2) * SED_MAX_GRU), ("gru_bhh", (_fp * 2) * SED_MAX_GRU),
        ("dense_w", _fp * SED_MAX_DENSE), ("dense_b", _fp * SED_MAX_DENSE),
    ]


_i, _l, _f, _d, _u64, _sz = C.c_int, C.c_long, C.c_float, C.c_double, C.c_uint64, C.c_size_t
_pp = C.POINTER(_fp)

# name -> (restype, argtypes); must list EVERY symbol of include/sedcrnn.h (tests check this)
SIGNATURES = {
    "sed_version": (_i, []),
    "sed_last_error_string": (C.c_char_p, []),
    "sed_conv3x3_pack_weights": (_i, [_fp, _fp, _fp, _i, _i, _stream]),
    "sed_conv3x3_stat_rows": (_i, [_i, _i, _i, _i, _i, _i]),
    "sed_conv3x3_fwd": (_i, [_fp, _i, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _stream]),
    "sed_conv3x3_pack_weights_ex": (_i, [_fp, _fp, _fp, _i, _i, _i, _stream]),
    "sed_conv3x3_fwd_ex": (_i, [_fp, _i, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _i, _stream]),
    "sed_conv3x3_wgrad_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "sed_conv3x3_wgrad_zero_row_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "sed_conv3x3_wgrad": (_i, [_fp, _i, _fp, _fp, _fp, _i, _i, _i, _i, _i, _stream]),
    "sed_conv3x3_wgrad_ex": (_i, [_fp, _i, _fp, _fp, _fp, _i, _i, _i, _i, _i, _i, _stream]),
    "sed_conv3x3_bn_relu_pool_eval_supported": (_i, [_i, _i, _i, _i, _i]),
    "sed_conv3x3_pack_weights_bn_folded": (_i, [_fp, _fp, _fp, _fp, _fp, _fp, _f, _fp, _fp, _i, _i, _stream]),
    "sed_conv3x3_bn_relu_pool_eval": (_i, [_fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _stream]),
    "sed_conv3x3_dgrad_bnred_rows": (_i, [_i, _i, _i, _i, _i]),
    "sed_conv3x3_dgrad_bnred": (_i, [_fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _f, _i, _i, _i, _i, _i, _i, _i, _i, _i, _stream]),
    "sed_conv3x3_wino_rows": (_i, [_i, _i, _i, _i, _i]),
    "sed_conv3x3_wino_phase_ticks": (_i, [_fp]),
    "sed_conv3x3_wino_packed_floats": (_sz, [_i, _i]),
    "sed_conv3x3_wino_pack_weights": (_i, [_fp, _fp, _fp, _i, _i, _stream]),
    "sed_conv3x3_wino_fwd": (_i, [_fp, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _stream]),
    "sed_conv3x3_wino_dgrad_bnred": (_i, [_fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _f, _i, _i, _i, _i, _i, _i, _i, _i, _i, _stream]),
    "sed_conv3x3_wino_pack_weights_bn_folded": (_i, [_fp, _fp, _fp, _fp, _fp, _fp, _f, _fp, _fp, _i, _i, _stream]),
    "sed_conv3x3_wino_bn_relu_pool_eval": (_i, [_fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _stream]),
    "sed_conv3x3_wino_rg_rows": (_i, [_i, _i, _i, _i, _i, _i]),
    "sed_conv3x3_wino_dgrad_bnred_rg": (_i, [_fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _f, _fp, _i, _fp, _fp, _i, _i, _i, _i, _i, _stream]),
    "sed_conv3x3_dgrad_bnred_rg_rows": (_i, [_i, _i, _i, _i, _i, _i]),
    "sed_conv3x3_dgrad_bnred_rg": (_i, [_fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _f, _fp, _i, _fp, _fp, _i, _i, _i, _i, _i, _stream]),
    "sed_conv1_fused_supported": (_i, [_i, _i, _i, _i, _i, _i]),
    "sed_conv1_fused_rows": (_i, [_i, _i]),
    "sed_conv1_stats_workspace_bytes": (_sz, [_i, _i, _i]),
    "sed_conv1_moments_doubles": (_sz, [_i]),
    "sed_conv1_stats": (_i, [_fp, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _fp, _stream]),
    "sed_conv1_bn_relu_pool_drop_fwd": (_i, [_fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _i, _i, _f, _u64, _fp, _fp, _stream]),
    "sed_conv1_rgrad_supported": (_i, [_i, _i, _i, _i, _i, _i]),
    "sed_conv1_bwd_wgrad_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "sed_conv1_bwd_wgrad": (_i, [_fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _f, _fp, _fp, _fp, _stream]),
    "sed_conv1_bwd_wgrad_assemble": (_i, [_fp, _i, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _fp, _fp, _fp, _stream]),
    "sed_conv1_route": (_i, [_fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _i, _i, _stream]),
    "sed_conv1_bwd_reduce": (_i, [_fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _i, _i, _f, _u64, _fp, _stream]),
    "sed_conv1_bwd_apply_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "sed_conv1_bwd_apply_wgrad": (_i, [_fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _i, _i, _f, _u64, _fp, _fp, _fp, _fp, _stream]),
    "sed_bn_finalize_train": (_i, [_fp, _i, _i, _d, _fp, _fp, _fp, _fp, _f, _f, _fp, _fp, _fp, _fp, _stream]),
    "sed_bn_stat_sums": (_i, [_fp, _i, _i, _fp, _stream]),
    "sed_bn_finalize_from_sums": (_i, [_fp, _i, _d, _fp, _fp, _fp, _fp, _f, _f, _fp, _fp, _fp, _fp, _stream]),
    "sed_bn_finalize_eval": (_i, [_fp, _fp, _fp, _fp, _f, _i, _fp, _fp, _stream]),
    "sed_bn_relu_pool_drop_fwd": (_i, [_fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _i, _i, _f, _u64, _fp, _stream]),
    "sed_bn_relu_pool_route": (_i, [_fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _i, _stream]),
    "sed_bn_bwd_rows": (_i, [_i, _i, _i]),
    "sed_bn_relu_pool_drop_bwd_reduce": (_i, [_fp, _fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _i, _i, _f, _u64, _fp, _stream]),
    "sed_bn_bwd_finalize": (_i, [_fp, _i, _i, _fp, _fp, _fp, _fp, _stream]),
    "sed_bn_bwd_finalize_small_gamma": (_i, [_fp, _i, _i, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _f, _stream]),
    "sed_bn_bwd_reduce_pooled_supported": (_i, [_i, _i, _i, _i, _i]),
    "sed_bn_bwd_reduce_pooled": (_i, [_fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _i, _i, _f, _stream]),
    "sed_bn_relu_pool_drop_bwd_apply": (_i, [_fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _i, _i, _f, _u64, _fp, _stream]),
    "sed_reduce_rows": (_i, [_fp, _i, _i, _i, _fp, _stream]),
    "sed_gemm_f32": (_i, [_fp, _l, _l, _fp, _l, _l, _fp, _l, _fp, _f, _i, _i, _i, _stream]),
    "sed_gemm_f32_workspace_bytes": (_sz, [_i, _i, _i]),
    "sed_gemm_f32_ws": (_i, [_fp, _l, _l, _fp, _l, _l, _fp, _l, _fp, _i, _i, _i, _fp, _stream]),
    "sed_gemm_f32_wgrad": (_i, [_fp, _l, _l, _fp, _l, _l, _fp, _l, _i, _i, _i, _fp, _stream]),
    "sed_linear_fwd": (_i, [_fp, _fp, _fp, _fp, _i, _i, _i, _i, _stream]),
    "sed_linear_bwd_workspace_bytes": (_sz, [_i, _i, _i]),
    "sed_linear_bwd": (_i, [_fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _stream]),
    "sed_gru_seq_workspace_bytes": (_sz, [_i]),
    "sed_gru_seq_fwd": (_i, [_fp, _pp, _pp, _fp, _fp, _fp, _i, _i, _i, _stream]),
    "sed_gru_seq_bwd_workspace_bytes": (_sz, [_i, _i]),
    "sed_gru_seq_bwd": (_i, [_fp, _fp, _pp, _fp, _fp, _pp, _pp, _fp, _i, _i, _i, _stream]),
    "sed_loss_fwd_bwd": (_i, [_fp, _fp, _i, _i, _f, _f, _i, _fp, _fp, _fp, _stream]),
    "sed_sigmoid": (_i, [_fp, _fp, _i, _stream]),
    "sed_scale": (_i, [_fp, _l, _f, _stream]),
    "sed_sqnorm_workspace_bytes": (_sz, [_l]),
    "sed_grad_norm_clip_coef": (_i, [_fp, _l, _f, _fp, _fp, _stream]),
    "sed_adam_step": (_i, [_fp, _fp, _fp, _fp, _l, _f, _f, _f, _f, _f, _i, _fp, _fp, _stream]),
    "sed_step_advance": (_i, [_fp, _stream]),
    "sed_logmel_tables_bytes": (_sz, [_fp, _i, _i]),
    "sed_logmel_build_tables": (_i, [_fp, _fp, _i, _i, _fp, _sz]),
    "sed_logmel": (_i, [_fp, _l, _fp, _sz, _fp, _fp, _fp, _i, _i, _i, _i, _stream]),
    "sed_window_batch": (_i, [_fp, _fp, _l, _i, _i, _i, _fp, _fp, _fp, _i, _i, _i, _fp, _fp, _i, _i, _i, _stream]),
    "sed_pack_sequences": (_i, [_fp, _l, _i, _i, _i, _i, _fp, _stream]),
    "sed_col_mean_std_workspace_bytes": (_sz, [_i]),
    "sed_col_mean_std": (_i, [_fp, _l, _i, _fp, _fp, _fp, _stream]),
    "sed_col_standardize": (_i, [_fp, _l, _i, _fp, _fp, _fp, _stream]),
    "sed_segment_counts": (_i, [_fp, _fp, _l, _i, _i, _f, _fp, _stream]),
    "sed_prof_enable": (_i, [C.c_uint]),
    "sed_prof_read": (_i, [_i, C.POINTER(_d), C.POINTER(_l), C.POINTER(_d)]),
    "sed_prof_tag_name": (C.c_char_p, [_i]),
    "sed_prof_tag_count": (_i, []),
    "sed_net_out_shape": (_i, [C.POINTER(NetCfg), C.POINTER(_i), C.POINTER(_i)]),
    "sed_net_workspace_bytes": (_sz, [C.POINTER(NetCfg), _i]),
    "sed_net_forward": (_i, [C.POINTER(NetCfg), C.POINTER(NetParams), _fp, _fp, _fp, _i, _u64, _fp, _stream]),
    "sed_net_sync_region": (_i, [C.POINTER(NetCfg), _i, _i, C.POINTER(_sz), C.POINTER(_sz)]),
    "sed_net_workspace_region": (_i, [C.POINTER(NetCfg), _i, C.c_char_p, _i, C.POINTER(_sz), C.POINTER(_sz)]),
    "sed_net_routing": (_i, [C.POINTER(NetCfg), C.POINTER(NetParams), _fp, _fp, _i, _fp, _stream]),
    "sed_net_forward_phases": (_i, [C.POINTER(NetCfg), C.POINTER(NetParams), _fp, _fp, _fp, _i, _u64, _i, _i, _f, _stream]),
    "sed_net_backward_phases": (_i, [C.POINTER(NetCfg), C.POINTER(NetParams), C.POINTER(NetParams), _fp, _fp, _fp, _u64, _i, _i, _f, _stream]),
    "sed_net_backward": (_i, [C.POINTER(NetCfg), C.POINTER(NetParams), C.POINTER(NetParams), _fp, _fp, _fp, _u64, _fp, _i, _i, _stream, _stream]),
    "sed_net_backward_ready_stage": (_i, [C.POINTER(NetCfg), _i]),
}

_lib = None


class SedHipError(RuntimeError):
    pass


def lib():
    """Load libsedcrnn.so once; raise loudly when it is absent (no CPU / torch fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SedHipError(
                f"{LIB_PATH} is missing: build it with `python -m sed_crnn_amd.build` "
                "(hipcc --offload-arch=gfx950). sed_crnn_amd has no fallback path.")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)            # AttributeError if the ABI and this table drift apart
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().sed_last_error_string().decode("utf-8", "replace")
        raise SedHipError(f"{what or 'libsedcrnn'} failed (code {rc}): {msg}")


def ptr(t):
    """Device (or host) address of a torch tensor / None as c_void_p."""
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


def stream_ptr():
    """hipStream_t of torch's current stream (kernels are enqueued behind torch's own work)."""
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
