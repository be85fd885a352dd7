"""ctypes binding of the C ABI in include/hugectr_amd.h (libhugectr_amd.so, gfx950 only).

There is NO fallback: if the HIP library is missing or a symbol is absent, importing fails loudly.
Nothing here touches oracle/ (the CPU oracle is test infrastructure only).
"""
import ctypes
import os

# torch FIRST: the PyTorch-ROCm wheel bundles its own HIP runtime (torch/lib/libamdhip64.so) and
# libhugectr_amd.so names the system one.  Streams and device pointers cross this boundary, so both
# must be the SAME runtime instance: with torch's already in the process the loader resolves this
# library's libamdhip64.so.7 to it.  Loaded the other way round (a script whose first import is
# `hugectr`) the process ends up with two runtimes and the second one finds no device.
import torch  # noqa: F401,E402
from ctypes import (POINTER, Structure, c_char_p, c_float, c_int, c_int64, c_size_t,
                    c_uint64, c_void_p)

_HERE = os.path.dirname(os.path.abspath(__file__))
# HCTR_LIB_VARIANT=<name>: a kernel variant built next to the product (csrc/Makefile, VARIANT= /
# TAG=) and measured against it; unset = the product library.  A name that was not built fails
# like a missing product library does.
_VARIANT = os.environ.get("HCTR_LIB_VARIANT", "")
LIB_PATH = os.path.join(_HERE, f"libhugectr_amd{'_' + _VARIANT if _VARIANT else ''}.so")

# enums (include/hugectr_amd.h)
OPT_FTRL, OPT_ADAM, OPT_RMSPROP, OPT_ADAGRAD, OPT_NESTEROV, OPT_MOMENTUM_SGD, OPT_SGD = range(7)
UPDATE_LOCAL, UPDATE_GLOBAL, UPDATE_LAZY_GLOBAL = range(3)
EMB_DISTRIBUTED, EMB_LOCALIZED = 0, 1
KEY_U32, KEY_I64 = 0, 1
F32, F16, BF16 = 0, 1, 2


class EmbeddingParams(Structure):
    """hctr_embedding_params"""
    _fields_ = [
        ("embedding_type", c_int), ("key_type", c_int), ("out_dtype", c_int),
        ("train_batch_size", c_size_t), ("evaluate_batch_size", c_size_t),
        ("max_vocabulary_size_per_gpu", c_size_t), ("embedding_vec_size", c_size_t),
        ("max_feature_num", c_size_t), ("slot_num", c_size_t), ("combiner", c_int),
        ("slot_size_array", POINTER(c_size_t)),
        ("optimizer", c_int), ("update_type", c_int), ("lr", c_float),
        ("beta1", c_float), ("beta2", c_float), ("epsilon", c_float),
        ("initial_accu_value", c_float), ("momentum_factor", c_float), ("atomic_update", c_int),
        ("scaler", c_float), ("rank", c_int), ("world", c_int), ("seed", c_uint64),
    ]


class DetOptParams(Structure):
    """hctr_det_opt_params"""
    _fields_ = [
        ("optimizer", c_int), ("lr", c_float), ("beta1", c_float), ("beta2", c_float),
        ("epsilon", c_float), ("momentum_factor", c_float), ("rmsprop_beta", c_float),
        ("ftrl_lambda1", c_float), ("ftrl_lambda2", c_float), ("ftrl_beta", c_float),
        ("scaler", c_float),
    ]


_P = c_void_p
_SZP = POINTER(c_size_t)
_SIGNATURES = {
    # name: (restype, argtypes)
    "hctr_last_error": (c_char_p, []),
    "hctr_version": (c_int, []),
    "hctr_hash_keys": (c_int, [_P, c_int, c_size_t, _P, _P]),
    "hctr_ht_create": (c_int, [c_size_t, c_int, POINTER(_P)]),
    "hctr_ht_destroy": (c_int, [_P]),
    "hctr_ht_clear": (c_int, [_P, _P]),
    "hctr_ht_get_insert": (c_int, [_P, _P, c_size_t, _P, _P, _P]),
    "hctr_ht_get_mark": (c_int, [_P, _P, c_size_t, _P, _P, _P]),
    "hctr_ht_insert": (c_int, [_P, _P, _P, c_size_t, _P]),
    "hctr_ht_size": (c_int, [_P, _P, POINTER(c_size_t)]),
    "hctr_ht_value_head": (c_int, [_P, _P, POINTER(c_size_t)]),
    "hctr_ht_set_value_head": (c_int, [_P, c_size_t, _P]),
    "hctr_ht_table_size": (c_size_t, [_P]),
    "hctr_ht_dump": (c_int, [_P, _P, _P, POINTER(c_size_t), _P]),
    "hctr_ht_error_flags": (c_int, [_P, _P, POINTER(ctypes.c_uint32)]),
    "hctr_ht_recover": (c_int, [_P, _P, c_size_t, _P]),
    "hctr_forward_pool": (c_int, [c_size_t, c_int, c_int, _P, c_int, _P, _P, _P, c_int, _P]),
    "hctr_forward_pool_weighted": (c_int, [c_size_t, c_int, c_int, _P, _P, _P, _P, _P, _P]),
    "hctr_expand_key_grads": (c_int, [c_size_t, c_int, c_int, _P, _P, _P, _P, _P]),
    "hctr_forward_pool_multihot": (c_int, [c_size_t, c_int, c_int, _P, c_int, _P, _P, _P, c_int, _P]),
    "hctr_forward_pool_mapped": (c_int, [c_size_t, c_int, c_int, _P, c_int, _P, _P, _P, c_int, c_int,
                                         c_size_t, c_size_t, _P, _P]),
    "hctr_ebc_route_whole": (c_int, [c_size_t, c_int, _P, _P, _P, c_int, _P, _P, _P, _P, c_size_t,
                                     _P]),
    "hctr_forward_reorder": (c_int, [c_size_t, c_int, c_int, c_int, _P, _P, c_int, _P]),
    "hctr_backward_reorder": (c_int, [c_size_t, c_int, c_int, c_int, _P, _P, c_int, _P]),
    "hctr_emb_create": (c_int, [POINTER(EmbeddingParams), POINTER(_P)]),
    "hctr_emb_destroy": (c_int, [_P]),
    "hctr_emb_init_params": (c_int, [_P, _P]),
    "hctr_emb_forward": (c_int, [_P, c_int, _P, _P, c_size_t, _P, _P]),
    "hctr_emb_forward_scale": (c_int, [_P, c_int, _P, _P]),
    "hctr_emb_poll_overflow": (c_int, [_P, _P]),
    "hctr_emb_backward": (c_int, [_P, _P, _P]),
    "hctr_emb_get_wgrad": (c_int, [_P, _P, _P]),
    "hctr_emb_update_params": (c_int, [_P, _P]),
    "hctr_emb_set_learning_rate": (c_int, [_P, c_float]),
    "hctr_emb_get_vocabulary_size": (c_int, [_P, _P, POINTER(c_size_t)]),
    "hctr_emb_get_max_vocabulary_size": (c_size_t, [_P]),
    "hctr_emb_slots_on_rank": (c_size_t, [_P]),
    "hctr_emb_check_overflow": (c_int, [_P, _P]),
    "hctr_emb_dump": (c_int, [_P, _P, _P, _P, POINTER(c_size_t), _P]),
    "hctr_emb_load": (c_int, [_P, _P, _P, _P, c_size_t, _P]),
    "hctr_emb_table_ptr": (_P, [_P]),
    "hctr_emb_opt_state_ptr": (_P, [_P, c_int]),
    "hctr_emb_value_index_ptr": (_P, [_P]),
    "hctr_emb_reset": (c_int, [_P, _P]),
    "hctr_emb_profiling": (c_int, [_P, c_int]),
    "hctr_emb_profile_get": (c_int, [_P, c_int, POINTER(ctypes.c_double), POINTER(c_uint64)]),
    "hctr_ebc_keys_to_indices": (c_int, [_P, c_int, c_size_t, c_int64, c_int, _P, _P]),
    "hctr_ebc_route_workspace_bytes": (c_size_t, [c_size_t, c_int]),
    "hctr_ebc_route_keys": (c_int, [c_size_t, c_int, c_int, _P, _P, _P, _P, c_int, _P, _P, _P, _P, _P]),
    "hctr_ebc_bucket_counts": (c_int, [c_size_t, c_int, c_int, c_int, _P, c_int, _P, _P]),
    "hctr_ebc_network_forward": (c_int, [c_size_t, c_int, c_int, c_int, _P, _P, _P, c_int, _P, _P, c_int, _P]),
    "hctr_ebc_network_backward": (c_int, [c_size_t, c_int, c_int, c_int, _P, _P, _P, c_int, _P, _P, c_int, _P]),
    "hctr_updater_create": (c_int, [c_size_t, c_size_t, c_int, POINTER(_P)]),
    "hctr_updater_destroy": (c_int, [_P]),
    "hctr_updater_update": (c_int, [_P, c_size_t, c_size_t, _P, _P, _P, c_int, c_int, c_int, c_float,
                                    c_float, c_float, c_float, c_float, c_float, c_uint64, _P, _P,
                                    _P, _P]),
    "hctr_interaction_fwd": (c_int, [c_size_t, c_int, c_int, _P, _P, _P, c_int, _P]),
    "hctr_interaction_bwd": (c_int, [c_size_t, c_int, c_int, _P, _P, _P, _P, _P, c_int, _P]),
    "hctr_cross_v1_fwd": (c_int, [c_size_t, c_int, c_int, _P, _P, _P, _P, _P, _P]),
    "hctr_cross_v1_bwd": (c_int, [c_size_t, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "hctr_cross_v1_bwd_workspace_bytes": (c_size_t, [c_size_t, c_int, c_int]),
    "hctr_det_create": (c_int, [c_size_t, _SZP, c_char_p, c_size_t, c_int, c_uint64, POINTER(_P)]),
    "hctr_det_destroy": (c_int, [_P]),
    "hctr_det_num_classes": (c_size_t, [_P]),
    "hctr_det_lookup": (c_int, [_P, _P, _P, c_size_t, _SZP, _SZP, c_size_t, _P]),
    "hctr_det_lookup_unsafe": (c_int, [_P, _P, _P, c_size_t, _SZP, _SZP, c_size_t, _P]),
    "hctr_det_scatter_add": (c_int, [_P, _P, _P, c_size_t, _SZP, _SZP, c_size_t, _P]),
    "hctr_det_scatter_update": (c_int, [_P, _P, _P, c_size_t, _SZP, _SZP, c_size_t, _P]),
    "hctr_det_remove": (c_int, [_P, _P, c_size_t, _SZP, _SZP, c_size_t, _P]),
    "hctr_det_export": (c_int, [_P, c_size_t, _P, _P, c_size_t, _SZP, _P]),
    "hctr_det_lookup_index": (c_int, [_P, c_size_t, _P, c_size_t, c_int, _P, _P]),
    "hctr_det_rows": (c_int, [_P, c_size_t, POINTER(_P), _SZP]),
    "hctr_det_row_store": (c_int, [_P, POINTER(_P), POINTER(ctypes.c_uint64)]),
    "hctr_det_state_store": (c_int, [_P, c_int, POINTER(_P), POINTER(_P), _P]),
    "hctr_det_lookup_rows": (c_int, [_P, _P, c_size_t, _SZP, _SZP, c_size_t, c_int, _P, _P,
                                     POINTER(c_uint64), _P]),
    "hctr_radix_sort_temp_bytes": (c_size_t, [c_size_t]),
    "hctr_radix_sort_pairs_u32": (c_int, [_P, c_size_t, _P, _P, _P, _P, c_size_t, c_int, _P]),
    "hctr_static_lookup": (c_int, [_P, c_int, c_size_t, _P, c_size_t, _P, _P, c_size_t, _P, _P, _P,
                                   _P, _P, _P, _P]),
    "hctr_forward_pool_ptrs": (c_int, [c_size_t, c_int, c_int, _P, _P, _P, c_int, _P]),
    "hctr_forward_pool_ptrs_mapped": (c_int, [c_size_t, c_int, c_int, _P, _P, _P, c_int, c_size_t,
                                              c_size_t, _P]),
    "hctr_ebc_routed_keys_to_indices": (c_int, [c_size_t, c_int, c_int, _P, _P, _P, _P, _P]),
    "hctr_ebc_local_reduce": (c_int, [_P, c_size_t, c_size_t, _P, _P, c_uint64, _P, _P, c_int, _SZP,
                                      _P, _P, _P, _P]),
    "hctr_det_clear": (c_int, [_P, _P]),
    "hctr_det_size_per_class": (c_int, [_P, _SZP, _P]),
    "hctr_det_capacity_per_class": (c_int, [_P, _SZP]),
    "hctr_det_repair_count": (c_int, [_P, POINTER(ctypes.c_uint64)]),
    "hctr_det_update": (c_int, [_P, _P, POINTER(DetOptParams), _P, c_size_t, _SZP, _SZP, c_size_t,
                                _P, _P, _P]),
    "hctr_cache_create": (c_int, [c_size_t, c_int, c_int, POINTER(_P)]),
    "hctr_cache_destroy": (c_int, [_P]),
    "hctr_cache_capacity_in_set": (c_size_t, [_P]),
    "hctr_cache_query": (c_int, [_P, _P, c_size_t, _P, _P, _P, _P, _P]),
    "hctr_cache_replace": (c_int, [_P, _P, c_size_t, _P, _P]),
    "hctr_cache_update": (c_int, [_P, _P, c_size_t, _P, _P]),
    "hctr_cache_dump": (c_int, [_P, _P, _P, c_size_t, c_size_t, _P]),
    "hctr_tiered_create": (c_int, [c_size_t, c_int, c_size_t, POINTER(_P)]),
    "hctr_tiered_destroy": (c_int, [_P]),
    "hctr_tiered_host_rows": (_P, [_P]),
    "hctr_tiered_cache": (_P, [_P]),
    "hctr_tiered_lookup": (c_int, [_P, _P, c_size_t, _P, _P, _P]),
    "hctr_tiered_scatter": (c_int, [_P, _P, c_size_t, _P, c_int, c_float, _P]),
    "hctr_tiered_flush": (c_int, [_P, _P]),
    "hctr_uvm_create": (c_int, [c_size_t, c_size_t, c_size_t, c_int, c_float, c_int, POINTER(_P)]),
    "hctr_uvm_destroy": (c_int, [_P]),
    "hctr_uvm_tier": (_P, [_P]),
    "hctr_uvm_add": (c_int, [_P, _P, _P, c_size_t]),
    "hctr_uvm_query": (c_int, [_P, _P, c_size_t, _P, _P]),
    "hctr_uvm_clear": (c_int, [_P, _P]),
    "hctr_uvm_lookup": (c_int, [_P, _P, c_size_t, _P, _P, _P, _P]),
    "hctr_uvm_scatter_rows": (c_int, [_P, _P, c_size_t, _P, c_int, c_float, _P]),
    "hctr_uvm_check_overflow": (c_int, [_P, _P]),
    "hctr_uvm_size": (c_int, [_P, _P, POINTER(c_size_t)]),
    "hctr_uniq_create": (c_int, [c_size_t, POINTER(_P)]),
    "hctr_uniq_destroy": (c_int, [_P]),
    "hctr_uniq_plan": (c_int, [_P, c_size_t, c_size_t, c_int, c_int, c_int, c_int, c_int, _P,
                               c_uint64, _P, _P, _P, _P]),
    "hctr_uniq_gather_rows": (c_int, [c_size_t, c_int, _P, _P, _P, c_int, _P]),
    "hctr_uniq_expand": (c_int, [c_size_t, c_int, _P, _P, _P, _P, c_int, c_int, _P, _P, _P, _P, _P]),
    "hctr_interaction_fwd_indexed": (c_int, [c_size_t, c_int, c_int, _P, _P, _P, _P, c_int, _P]),
    "hctr_ebc_scale_average": (c_int, [c_size_t, c_int, c_int, _P, _P, c_int, _P, c_int, c_int, _P]),
    "hctr_interaction_fwd_gather": (c_int, [c_size_t, c_int, c_int, _P, _P, _P, _P, _P, c_int, _P]),
    "hctr_emb_forward_interaction": (c_int, [_P, c_int, _P, _P, _P, _P]),
    "hctr_interaction_bwd_indexed": (c_int, [c_size_t, c_int, c_int, _P, _P, _P, _P, _P, _P, c_int,
                                             _P]),
    "hctr_interaction_bwd_indexed_scatter": (c_int, [c_size_t, c_int, c_int, _P, _P, _P, _P, _P, _P,
                                                     c_int, _P]),
    "hctr_updater_set_ftrl": (c_int, [_P, c_float, c_float, c_float]),
    "hctr_updater_set_grad_map": (c_int, [_P, c_size_t, c_size_t]),
    "hctr_updater_set_row_bound": (c_int, [_P, ctypes.c_uint64]),
    "hctr_updater_reduce_presorted": (c_int, [_P, c_size_t, c_size_t, _P, _P, _P, _P, c_int,
                                              c_size_t, _P, _P]),
    "hctr_emb_index": (c_int, [_P, c_int, _P, _P, c_size_t, _P]),
    "hctr_emb_index_ahead": (c_int, [_P, _P, _P, c_size_t, _P]),
    "hctr_emb_index_adopt": (c_int, [_P]),
    "hctr_emb_update_rows": (c_int, [_P, c_size_t, _P, _P, _P, c_int, _P]),
    "hctr_relu_bwd_bias_workspace_bytes": (c_size_t, [c_size_t, c_int]),
    "hctr_relu_bwd_bias": (c_int, [c_size_t, c_int, _P, _P, _P, _P, _P, c_int, _P]),
    "hctr_logit_head_workspace_bytes": (c_size_t, [c_int]),
    "hctr_logit_head": (c_int, [c_size_t, c_int, _P, _P, _P, _P, c_float, _P, _P, _P, _P, _P, c_int,
                                _P]),
    "hctr_skinny_fc_fwd": (c_int, [c_size_t, c_int, c_int, _P, _P, _P, _P, c_int, _P]),
    "hctr_skinny_fc_bwd_workspace_bytes": (c_size_t, [c_int]),
    "hctr_skinny_fc_bwd": (c_int, [c_size_t, c_int, c_int, _P, _P, _P, _P, _P, _P, c_int, _P]),
    "hctr_sum_groups": (c_int, [c_int, c_size_t, _P, c_int, _P, _P]),
    "hctr_sgd_shadow": (c_int, [c_size_t, c_float, c_float, _P, _P, _P, c_int, _P]),
    "hctr_bce_loss_workspace_bytes": (c_size_t, []),
    "hctr_bce_loss": (c_int, [c_size_t, _P, _P, c_float, _P, _P, _P, c_int, _P]),
    "hctr_convert_transpose16": (c_int, [c_size_t, c_int, c_int, _P, _P, _P, c_int, _P]),
    "hctr_gemm_nt16": (c_int, [c_size_t, c_int, c_int, _P, c_int, _P, c_int, _P, c_int, c_int, _P, _P,
                               _P, _P, c_int, _P]),
    "hctr_cross_v2_bwd_step_workspace_bytes": (c_size_t, [c_size_t, c_int]),
    "hctr_cross_v2_bwd_step": (c_int, [c_size_t, c_int, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, _P]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES.keys())


class HugeCTRAmdError(RuntimeError):
    """The reference surfaces C++ exceptions as Python RuntimeError; so do we."""


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing: fail loudly
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def last_error() -> str:
    msg = lib.hctr_last_error()
    return msg.decode() if msg else ""


def check(rc: int) -> None:
    if rc != 0:
        raise HugeCTRAmdError(f"hugectr_amd error {rc}: {last_error()}")


def ptr(t):
    """device pointer of a torch tensor (or None)."""
    if t is None:
        return None
    return c_void_p(t.data_ptr())


def stream_ptr():
    """current torch HIP stream as hipStream_t."""
    import torch
    return c_void_p(torch.cuda.current_stream().cuda_stream)
