"""ctypes binding of libvqa_mi355x.so (C ABI: include/vqa_mi355x.h).

The product path has NO fallback: if the shared library is missing or a symbol is absent,
``lib()`` raises.  Build it with ``python __graft_entry__.py build`` (hipcc, gfx950).
"""
import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VQA_LIB_PATH", os.path.join(_HERE, "libvqa_mi355x.so"))  # env override: profiling builds
ABI_VERSION = 13

_c_f = ctypes.c_void_p          # device pointer to fp32
_c_pp = ctypes.c_void_p         # host array of device pointers
_c_i = ctypes.c_int
_c_sz = ctypes.c_size_t
_c_u64 = ctypes.c_uint64
_c_l = ctypes.c_long
_c_fl = ctypes.c_float
_c_st = ctypes.c_void_p         # hipStream_t

# name -> (restype, argtypes); every symbol include/vqa_mi355x.h declares
SIGNATURES = {
    "vqa_version": (_c_i, []),
    "vqa_last_error": (ctypes.c_char_p, []),
    "vqa_source_hash": (ctypes.c_char_p, []),
    "vqa_set_option": (_c_i, [ctypes.c_char_p, ctypes.c_char_p]),
    "vqa_launch_log_reset": (None, []),
    "vqa_launch_log": (_c_i, [ctypes.POINTER(ctypes.c_ulonglong), _c_i]),
    "vqa_launch_log_kernel": (ctypes.c_char_p, [_c_i]),
    "vqa_pairwise_relation_reduce_fwd": (_c_i, [_c_f, _c_f, _c_f, _c_f, _c_i, _c_f, _c_i, _c_i, _c_i, _c_i, _c_st]),
    "vqa_pairwise_relation_reduce_bwd": (_c_i, [_c_f, _c_f, _c_f, _c_f, _c_i, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f,
                                                _c_i, _c_i, _c_i, _c_st]),
    "vqa_relation_apply_fwd": (_c_i, [_c_f, _c_f, _c_f, _c_f, _c_fl, _c_u64, _c_f, _c_i, _c_i, _c_i, _c_st]),
    "vqa_relation_apply_bwd": (_c_i, [_c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_fl, _c_u64, _c_f, _c_i, _c_i, _c_i, _c_st]),
    "vqa_relation_apply_fwd_bf16": (_c_i, [_c_f, _c_f, _c_f, _c_f, _c_fl, _c_u64, _c_f, _c_i, _c_i, _c_i, _c_st]),
    "vqa_relation_apply_bwd_bf16": (_c_i, [_c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_fl, _c_u64, _c_f, _c_i, _c_i, _c_i, _c_st]),
    "vqa_softmax_attention_pool_fwd": (_c_i, [_c_f, _c_f, _c_f, _c_f, _c_i, _c_i, _c_i, _c_i, _c_st]),
    "vqa_softmax_attention_pool_bwd": (_c_i, [_c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_i, _c_i, _c_i, _c_i, _c_st]),
    "vqa_attention_logits_fwd": (_c_i, [_c_f, _c_i, _c_f, _c_f, _c_f, _c_fl, _c_u64, _c_f, _c_i, _c_i, _c_i, _c_st]),
    "vqa_attention_logits_fwd_bf16": (_c_i, [_c_f, _c_i, _c_f, _c_f, _c_f, _c_fl, _c_u64, _c_f, _c_i, _c_i, _c_i, _c_st]),
    "vqa_attention_logits_bwd_workspace_bytes": (_c_sz, [_c_i, _c_i, _c_i]),
    "vqa_attention_logits_bwd": (_c_i, [_c_f, _c_i, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_sz, _c_fl, _c_u64, _c_f,
                                        _c_i, _c_i, _c_i, _c_st]),
    "vqa_attention_logits_bwd_bf16": (_c_i, [_c_f, _c_i, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_sz, _c_fl, _c_u64, _c_f,
                                             _c_i, _c_i, _c_i, _c_st]),
    "vqa_lowrank_bilinear_fusion_fwd": (_c_i, [_c_f, _c_i, _c_pp, _c_pp, _c_f, _c_f, _c_f,
                                               _c_i, _c_i, _c_i, _c_i, _c_i, _c_st]),
    "vqa_lowrank_bilinear_fusion_bwd_workspace_bytes": (_c_sz, [_c_i, _c_i, _c_i, _c_i, _c_i]),
    "vqa_lowrank_bilinear_fusion_bwd": (_c_i, [_c_f, _c_i, _c_pp, _c_f, _c_f, _c_f, _c_f, _c_pp, _c_pp, _c_f,
                                               _c_f, _c_sz, _c_i, _c_i, _c_i, _c_i, _c_i, _c_st]),
    "vqa_lowrank_bilinear_fusion_folded_supported": (_c_i, [_c_i, _c_i, _c_i, _c_i, _c_i]),
    "vqa_lowrank_bilinear_fusion_folded_fwd": (_c_i, [_c_f, _c_i, _c_pp, _c_pp, _c_f, _c_f,
                                                      _c_i, _c_i, _c_i, _c_i, _c_i, _c_st]),
    "vqa_lowrank_bilinear_fusion_folded_bwd_workspace_bytes": (_c_sz, [_c_i, _c_i, _c_i, _c_i, _c_i]),
    "vqa_lowrank_bilinear_fusion_folded_bwd": (_c_i, [_c_f, _c_i, _c_pp, _c_pp, _c_f, _c_f, _c_f, _c_pp, _c_pp, _c_f,
                                                      _c_f, _c_sz, _c_i, _c_i, _c_i, _c_i, _c_i, _c_st]),
    "vqa_lowrank_bilinear_fusion_folded_bwd_gated": (_c_i, [_c_f, _c_i, _c_pp, _c_pp, _c_f, _c_f, _c_f, _c_pp, _c_pp, _c_f,
                                                            _c_f, _c_sz, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_st]),
    "vqa_dropout_groups_fwd": (_c_i, [_c_f, _c_i, _c_f, _c_fl, _c_u64, _c_f, _c_i, _c_i, _c_i, _c_st]),
    "vqa_dropout_groups_bwd": (_c_i, [_c_f, _c_f, _c_fl, _c_u64, _c_f, _c_i, _c_i, _c_i, _c_st]),
    "vqa_pairwise_relation_reduce_drop_supported": (_c_i, [_c_i, _c_i, _c_i]),
    "vqa_pairwise_relation_reduce_drop_fwd": (_c_i, [_c_f, _c_f, _c_f, _c_f, _c_i, _c_f, _c_fl, _c_u64, _c_f, _c_i, _c_i, _c_i, _c_st]),
    "vqa_relation_projection_dgrad_supported": (_c_i, [_c_i, _c_i, _c_i, _c_i]),
    "vqa_relation_projection_dgrad": (_c_i, [_c_f, _c_f, _c_f, _c_f, _c_f, _c_fl, _c_u64, _c_f, _c_i, _c_i, _c_i, _c_i, _c_st]),
    "vqa_rank_product_fwd": (_c_i, [_c_f, _c_f, _c_f, _c_i, _c_i, _c_i, _c_st]),
    "vqa_rank_product_bwd": (_c_i, [_c_f, _c_f, _c_f, _c_f, _c_f, _c_i, _c_i, _c_i, _c_st]),
    "vqa_object_difference_attention_fwd": (_c_i, [_c_f, _c_f, _c_f, _c_f, _c_f, _c_fl, _c_u64, _c_f,
                                                   _c_i, _c_i, _c_i, _c_i, _c_st]),
    "vqa_object_difference_attention_bwd_workspace_bytes": (_c_sz, [_c_i, _c_i, _c_i, _c_i]),
    "vqa_object_difference_attention_bwd": (_c_i, [_c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_sz,
                                                   _c_fl, _c_u64, _c_f, _c_i, _c_i, _c_i, _c_i, _c_i, _c_st]),
    "vqa_object_difference_dropout_mask": (_c_i, [_c_f, _c_fl, _c_u64, _c_f, _c_i, _c_i, _c_i, _c_st]),
    "vqa_linear_act_fwd": (_c_i, [_c_f, _c_i, _c_f, _c_f, _c_f, _c_i, _c_i, _c_i, _c_i, _c_fl, _c_u64, _c_f, _c_st]),
    "vqa_linear_act_bwd_workspace_bytes": (_c_sz, [_c_i, _c_i, _c_i]),
    "vqa_linear_act_bwd": (_c_i, [_c_f, _c_i, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_sz,
                                  _c_i, _c_i, _c_i, _c_i, _c_fl, _c_u64, _c_f, _c_st]),
    "vqa_linear_dropout_mask": (_c_i, [_c_f, _c_fl, _c_u64, _c_f, _c_i, _c_i, _c_st]),
    "vqa_linear_split_supported": (_c_i, [_c_i, _c_i, _c_i, _c_i, _c_fl]),
    "vqa_relation_projection_dgrad_split_supported": (_c_i, [_c_i, _c_i, _c_i, _c_i]),
    "vqa_relation_projection_dgrad_split_workspace_bytes": (_c_sz, [_c_i, _c_i]),
    "vqa_relation_projection_dgrad_split": (_c_i, [_c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_sz, _c_fl, _c_u64, _c_f, _c_i, _c_i, _c_i,
                                                   _c_i, _c_st]),
    "vqa_linear_act_fwd_split_workspace_bytes": (_c_sz, [_c_i, _c_i]),
    "vqa_linear_act_fwd_split": (_c_i, [_c_f, _c_i, _c_f, _c_f, _c_f, _c_f, _c_sz, _c_i, _c_i, _c_i, _c_i, _c_fl, _c_u64, _c_f, _c_st]),
    "vqa_split_weights_bytes": (_c_sz, [_c_i, _c_i, _c_i]),
    "vqa_split_weights_pack": (_c_i, [_c_f, _c_l, _c_i, _c_i, _c_f, _c_sz, _c_i, _c_i, _c_i, _c_st]),
    "vqa_gemm_nt_split_batched_supported": (_c_i, [_c_i, _c_i, _c_i, _c_i, _c_i]),
    "vqa_gemm_nt_split_batched": (_c_i, [_c_f, _c_l, _c_i, _c_f, _c_f, _c_l, _c_i, _c_f, _c_i, _c_f, _c_l, _c_i, _c_i, _c_i, _c_i,
                                         _c_i, _c_i, _c_st]),
    "vqa_host_gather_rows": (_c_i, [_c_f, _c_l, _c_l, ctypes.c_void_p, _c_i, ctypes.c_void_p, _c_i, _c_i]),
    "vqa_widen_bf16": (_c_i, [_c_f, _c_f, _c_sz, _c_st]),
    "vqa_gemm_tn_split_supported": (_c_i, [_c_i, _c_i, _c_i, _c_i, _c_i]),
    "vqa_gemm_tn_split_workspace_bytes": (_c_sz, [_c_i, _c_i, _c_i]),
    "vqa_gemm_tn_split": (_c_i, [_c_f, _c_i, _c_f, _c_i, _c_f, _c_f, _c_sz, _c_i, _c_i, _c_i, _c_st]),
    "vqa_relation_linear_split_supported": (_c_i, [_c_i, _c_i, _c_i, _c_i, _c_fl]),
    "vqa_relation_linear_fwd_split": (_c_i, [_c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_sz, _c_i, _c_i, _c_i, _c_i, _c_i, _c_fl,
                                             _c_u64, _c_f, _c_st]),
    "vqa_relation_linear_dw_split": (_c_i, [_c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_sz, _c_i, _c_i, _c_i, _c_i,
                                            _c_i, _c_fl, _c_u64, _c_f, _c_st]),
    "vqa_linear_act_dw_split_workspace_bytes": (_c_sz, [_c_i, _c_i, _c_i]),
    "vqa_linear_act_dw_split": (_c_i, [_c_f, _c_i, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_sz, _c_i, _c_i, _c_i, _c_i, _c_fl, _c_u64,
                                       _c_f, _c_st]),
    # bf16 (mixed-precision) side
    "vqa_pairwise_relation_reduce_fwd_bf16": (_c_i, [_c_f, _c_f, _c_f, _c_f, _c_i, _c_f, _c_i, _c_i, _c_i, _c_i, _c_st]),
    "vqa_pairwise_relation_reduce_bwd_bf16": (_c_i, [_c_f, _c_f, _c_f, _c_f, _c_i, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f,
                                                     _c_i, _c_i, _c_i, _c_st]),
    "vqa_softmax_attention_pool_drop_fwd": (_c_i, [_c_f, _c_f, _c_f, _c_f, _c_f, _c_fl, _c_u64, _c_f, _c_i, _c_i, _c_i, _c_i, _c_st]),
    "vqa_softmax_attention_pool_drop_bwd": (_c_i, [_c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_fl, _c_u64, _c_f, _c_i, _c_i, _c_i,
                                            _c_i, _c_st]),
    "vqa_softmax_attention_pool_drop_fwd_bf16": (_c_i, [_c_f, _c_f, _c_f, _c_f, _c_f, _c_fl, _c_u64, _c_f, _c_i, _c_i, _c_i, _c_i,
                                                 _c_st]),
    "vqa_softmax_attention_pool_drop_bwd_bf16": (_c_i, [_c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_fl, _c_u64, _c_f, _c_i, _c_i,
                                                 _c_i, _c_i, _c_st]),
    "vqa_softmax_attention_pool_fwd_bf16": (_c_i, [_c_f, _c_f, _c_f, _c_f, _c_i, _c_i, _c_i, _c_i, _c_st]),
    "vqa_softmax_attention_pool_bwd_bf16": (_c_i, [_c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_i, _c_i, _c_i, _c_i, _c_st]),
    "vqa_pack_bf16": (_c_i, [_c_f, _c_i, _c_i, _c_i, _c_f, _c_l, _c_l, _c_l, _c_sz, _c_i, _c_st]),
    "vqa_gemm_bf16_nt": (_c_i, [_c_f, _c_i, _c_f, _c_i, _c_f, _c_f, _c_i, _c_i, _c_i, _c_i, _c_i, _c_st]),
    "vqa_pack_many": (_c_i, [_c_f, _c_i, _c_sz, _c_st]),
    "vqa_gemm_bf16_nt_ex": (_c_i, [_c_f, _c_i, _c_f, _c_i, _c_f, _c_f, _c_i, _c_i, _c_i, _c_i, _c_i, _c_f, _c_i,
                                   _c_fl, _c_u64, _c_f, _c_st]),
    "vqa_gemm_bf16_tn_ex": (_c_i, [_c_f, _c_i, _c_f, _c_i, _c_pp, _c_i, _c_i, _c_i, _c_i, _c_i, _c_f, _c_sz,
                                   _c_i, _c_i, _c_i, _c_fl, _c_u64, _c_f, _c_st]),
    "vqa_gemm_bf16_tn_workspace_bytes": (_c_sz, [_c_i, _c_i, _c_i]),
    "vqa_gemm_bf16_tn": (_c_i, [_c_f, _c_i, _c_f, _c_i, _c_f, _c_f, _c_sz, _c_i, _c_i, _c_i, _c_st]),
    "vqa_lowrank_bilinear_fusion_fwd_bf16": (_c_i, [_c_f, _c_f, _c_f, _c_f, _c_f, _c_f,
                                                    _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_st]),
    "vqa_lowrank_bilinear_fusion_bwd_bf16_workspace_bytes": (_c_sz, [_c_i, _c_i, _c_i, _c_i, _c_i]),
    "vqa_lowrank_bilinear_fusion_bwd_bf16": (_c_i, [_c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_pp, _c_pp, _c_f, _c_f, _c_sz,
                                                    _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_st]),
    "vqa_bilinear_fold_bf16_supported": (_c_i, [_c_i, _c_i, _c_i, _c_i, _c_i]),
    "vqa_bilinear_fold_fwd_bf16": (_c_i, [_c_f, _c_f, _c_f, _c_f, _c_f, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_st]),
    "vqa_bilinear_fold_bwd_bf16_workspace_bytes": (_c_sz, [_c_i, _c_i, _c_i, _c_i, _c_i]),
    "vqa_bilinear_fold_bwd_bf16": (_c_i, [_c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_pp, _c_pp, _c_f, _c_f, _c_sz,
                                          _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_st]),
    "vqa_gate_product_fwd": (_c_i, [_c_f, _c_f, ctypes.c_long, _c_f, _c_i, _c_i, _c_st]),
    "vqa_gate_product_bwd": (_c_i, [_c_f, _c_f, _c_f, _c_f, _c_f, _c_f, ctypes.c_long, _c_f, _c_f, _c_f, _c_i, _c_i, _c_st]),
    "vqa_grouped_gemm": (_c_i, [ctypes.c_void_p, _c_i, _c_st]),
    "vqa_grouped_gemm_split": (_c_i, [ctypes.c_void_p, _c_i, _c_st]),
    "vqa_grouped_gemm_split_tile_cols": (_c_i, [_c_i]),
    "vqa_grouped_epilogue": (_c_i, [ctypes.c_void_p, _c_i, _c_st]),
    "vqa_gru_gates_fwd": (_c_i, [_c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_sz, _c_f, _c_f, _c_f, _c_f,
                                 _c_i, _c_i, _c_i, _c_i, _c_i, _c_st]),
    "vqa_gru_gates_bwd": (_c_i, [_c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_sz, _c_f, _c_f,
                                 _c_i, _c_i, _c_i, _c_i, _c_i, _c_st]),
    "vqa_bias_act": (_c_i, [_c_f, _c_f, _c_i, _c_f, _c_i, _c_i, _c_i, _c_i, _c_i, _c_st]),
    "vqa_act_bwd_colsum": (_c_i, [_c_f, _c_f, _c_f, _c_f, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_st]),
    "vqa_column_sum_workspace_bytes": (_c_sz, [_c_i, _c_i]),
    "vqa_column_sum": (_c_i, [_c_f, _c_i, _c_f, _c_f, _c_sz, _c_i, _c_i, _c_st]),
    "vqa_column_sum_bf16": (_c_i, [_c_f, _c_i, _c_f, _c_f, _c_sz, _c_i, _c_i, _c_st]),
    "vqa_kld_sum_loss_workspace_bytes": (_c_sz, [_c_i]),
    "vqa_kld_sum_loss": (_c_i, [_c_f, _c_f, _c_f, _c_f, _c_f, _c_sz, _c_i, _c_i, _c_st]),
    "vqa_grad_norm_workspace_bytes": (_c_sz, []),
    "vqa_grad_norm_clip_coef": (_c_i, [_c_f, _c_sz, _c_fl, _c_f, _c_f, _c_sz, _c_st]),
    "vqa_adam_step": (_c_i, [_c_f, _c_f, _c_f, _c_f, _c_sz, _c_f, _c_fl, _c_fl, _c_fl, _c_fl, _c_i, _c_st]),
    "vqa_adam_step_dyn": (_c_i, [_c_f, _c_f, _c_f, _c_f, _c_sz, _c_f, _c_f, _c_fl, _c_fl, _c_fl, _c_st]),
}

_lock = threading.Lock()
_lib = None


class VqaLibraryError(RuntimeError):
    """libvqa_mi355x.so is missing, stale, or returned an error code."""


def lib():
    """Load (once) and return the ctypes handle.  Raises VqaLibraryError -- never falls back."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise VqaLibraryError(
                "HIP extension %s not found: build it with `python __graft_entry__.py build` "
                "(hipcc --offload-arch=gfx950).  There is no CPU or PyTorch fallback for this path." % LIB_PATH)
        try:
            handle = ctypes.CDLL(LIB_PATH)
        except OSError as e:  # pragma: no cover - depends on the host's ROCm install
            raise VqaLibraryError("cannot load %s: %s" % (LIB_PATH, e)) from e
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(handle, name)
            except AttributeError as e:
                raise VqaLibraryError("%s does not export %s (stale build?)" % (LIB_PATH, name)) from e
            fn.restype = res
            fn.argtypes = args
        if handle.vqa_version() != ABI_VERSION:
            raise VqaLibraryError("ABI version mismatch: library %d, binding %d" % (handle.vqa_version(), ABI_VERSION))
        _lib = handle
    return _lib


def set_option(name, value):
    """Override (str / int) or clear (None) one of the library's VQA_* knobs for the rest of the process.  The library
    reads each knob from the environment once, at its first use; this is the explicit way to change one afterwards
    (tests, tools) -- never between the forward and the backward of an op."""
    v = None if value is None else str(value).encode()
    check(lib().vqa_set_option(name.encode(), v), "set_option")


def check(rc, what):
    if rc != 0:
        msg = lib().vqa_last_error()
        raise VqaLibraryError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else "?"))


def built_from_this_tree():
    """(library's vqa_source_hash(), hash of the source tree beside it): equal when the loaded binary was built from the sources
    in vqa_playground_pytorch_amd/csrc + include/ as they are now.  __graft_entry__.smoke() and tests/test_host_cpu.py assert
    it, so a stale .so that travelled to the GPU box is seen there."""
    from . import _srchash
    return (lib().vqa_source_hash() or b"").decode(), _srchash.source_hash()
