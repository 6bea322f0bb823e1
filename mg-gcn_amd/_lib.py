"""ctypes binding of libmggcn_hip.so (the C ABI declared in include/mggcn.h).

The library is the product: every device computation of this package goes
through it.  There is no fallback -- if the shared object is missing or no
MI355X is visible, the first call that needs it raises ``engine_error``.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_float, c_int, c_int32, c_size_t, c_uint32, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libmggcn_hip.so")


class engine_error(RuntimeError):
    """The HIP engine is unusable (library not built / no GPU)."""


u32p = ctypes.POINTER(c_uint32)
f32p = ctypes.POINTER(c_float)
vp = c_void_p

# name -> (restype, argtypes); one entry per declaration in include/mggcn.h
PROTOTYPES = {
    "mggcn_abi_version": (c_int, []),
    "mggcn_device_count": (c_int, []),
    "mggcn_set_device": (None, [c_int]),
    "mggcn_get_device": (c_int, []),
    "mggcn_device_synchronize": (None, []),
    "mggcn_stream_create": (vp, [c_int]),
    "mggcn_stream_destroy": (None, [vp]),
    "mggcn_stream_release_scratch": (None, [vp]),
    "mggcn_stream_synchronize": (None, [vp]),
    "mggcn_event_create": (vp, []),
    "mggcn_event_destroy": (None, [vp]),
    "mggcn_event_record": (None, [vp, vp]),
    "mggcn_stream_wait_event": (None, [vp, vp]),
    "mggcn_event_synchronize": (None, [vp]),
    "mggcn_event_elapsed_ms": (c_float, [vp, vp]),
    "mggcn_malloc": (vp, [c_size_t]),
    "mggcn_free": (None, [vp]),
    "mggcn_malloc_host": (vp, [c_size_t]),
    "mggcn_free_host": (None, [vp]),
    "mggcn_memcpy_h2d": (None, [vp, vp, c_size_t, vp]),
    "mggcn_memcpy_d2h": (None, [vp, vp, c_size_t, vp]),
    "mggcn_memcpy_d2d": (None, [vp, vp, c_size_t, vp]),
    "mggcn_memset_zero": (None, [vp, c_size_t, vp]),
    "mggcn_spmm_plan_create": (vp, [c_uint32, c_uint32, vp, vp, vp, c_uint32]),
    "mggcn_spmm_plan_create_for": (vp, [c_uint32, c_uint32, vp, vp, vp, c_uint32, c_uint32]),
    "mggcn_spmm_plan_destroy": (None, [vp]),
    "mggcn_spmm_plan_concurrent_builders": (None, [c_uint32]),
    "mggcn_spmm_plan_reserved_cus": (None, [c_uint32]),
    "mggcn_debug_occupy_cus": (None, [vp, c_uint32, c_uint32, vp]),
    "mggcn_spmm_plan_num_items": (c_uint32, [vp]),
    "mggcn_spmm_plan_num_split_rows": (c_uint32, [vp]),
    "mggcn_spmm_plan_num_sweep_tasks": (c_uint32, [vp]),
    "mggcn_spmm_plan_num_launches": (c_uint32, [vp, c_uint32]),
    "mggcn_spmm_plan_bytes": (c_size_t, [vp]),
    "mggcn_spmm_plan_num_slices": (c_uint32, [vp]),
    "mggcn_spmm_plan_describe": (c_int, [vp, ctypes.c_char_p, c_size_t]),
    "mggcn_spmm_plan_read_stamps": (c_uint32, [vp, c_uint32, vp, c_uint32]),
    "mggcn_spmm_csr_f32": (None, [vp, vp, c_uint32, c_uint32, vp, vp, vp, vp, c_size_t, vp, c_size_t,
                                  c_uint32, c_float, c_float, c_uint32, c_float]),
    "mggcn_gemm_workspace_bytes": (c_size_t, [c_int, c_int, c_uint32, c_uint32, c_uint32]),
    "mggcn_gemm_f32": (None, [vp, c_int, c_int, c_uint32, c_uint32, c_uint32, c_float, vp, c_size_t, vp,
                              c_size_t, c_float, vp, c_size_t, vp, c_size_t]),
    "mggcn_gemm_bias_f32": (None, [vp, c_int, c_int, c_uint32, c_uint32, c_uint32, c_float, vp, c_size_t, vp,
                                   c_size_t, vp, vp, c_size_t, vp, c_size_t]),
    "mggcn_gemm_tn_colsum_workspace_bytes": (c_size_t, [c_uint32, c_uint32, c_uint32]),
    "mggcn_gemm_tn_colsum_f32": (None, [vp, c_uint32, c_uint32, c_uint32, c_float, vp, c_size_t, vp, c_size_t, vp,
                                        c_size_t, vp, vp, c_size_t]),
    "mggcn_gemm_lrelu_bwd_f32": (None, [vp, c_int, c_int, c_uint32, c_uint32, c_uint32, c_float, vp, c_size_t, vp,
                                        c_size_t, vp, c_size_t, c_float, vp, c_size_t, vp, c_size_t]),
    "mggcn_leaky_relu_forward_f32": (None, [vp, vp, vp, c_size_t, c_float]),
    "mggcn_leaky_relu_backward_f32": (None, [vp, vp, vp, vp, c_size_t, c_float]),
    "mggcn_broadcast_rows_f32": (None, [vp, vp, vp, c_size_t, c_size_t, c_int]),
    "mggcn_scale_rows_f32": (None, [vp, vp, vp, c_size_t, c_size_t]),
    "mggcn_max_rows_f32": (None, [vp, vp, vp, c_size_t, c_size_t]),
    "mggcn_max_row_indices_f32": (None, [vp, vp, vp, c_size_t, c_size_t]),
    "mggcn_index_log_rows_f32": (None, [vp, vp, vp, vp, c_size_t, c_size_t]),
    "mggcn_add_indexed_rows_f32": (None, [vp, vp, vp, c_float, c_size_t, c_size_t]),
    "mggcn_is_equal_i32": (None, [vp, vp, vp, vp, c_size_t]),
    "mggcn_subtract_rows_exp_f32": (None, [vp, vp, vp, vp, c_size_t, c_size_t]),
    "mggcn_axpby_f32": (None, [vp, vp, vp, c_float, c_float, c_size_t]),
    "mggcn_aaxpby_f32": (None, [vp, vp, vp, c_float, c_float, c_size_t]),
    "mggcn_adam_final_f32": (None, [vp, vp, vp, vp, c_float, c_float, c_float, c_float, c_size_t]),
    "mggcn_axpy_f32": (None, [vp, vp, vp, c_float, c_size_t]),
    "mggcn_scale_mat_f32": (None, [vp, vp, c_float, c_size_t]),
    "mggcn_abssum_f32": (None, [vp, vp, c_size_t, vp]),
    "mggcn_gather_rows_f32": (None, [vp, vp, c_size_t, vp, c_size_t, c_uint32, vp, c_size_t]),
    "mggcn_softmax_xent_fused_f32": (None, [vp, vp, vp, c_size_t, c_size_t, c_float, vp]),
    "mggcn_softmax_xent_fused_from_f32": (None, [vp, vp, vp, vp, c_size_t, c_size_t, c_float, vp]),
    "mggcn_adam_fused_f32": (None, [vp, vp, vp, vp, vp, c_float, c_float, c_float, c_float, c_float,
                                    c_float, c_float, c_size_t]),
    "mggcn_adam_multi_blocks": (c_uint32, [c_uint64]),
    "mggcn_adam_multi_f32": (None, [vp, vp, c_uint32, c_uint32, c_float, c_float, c_float, c_float, c_float, c_float]),
    "mggcn_csr_normalize_host": (None, [c_uint32, c_uint32, vp, vp, vp, c_int]),
    "mggcn_csr_transpose_host": (None, [c_uint32, c_uint32, vp, vp, vp, vp, vp, vp]),
    "mggcn_csr_block_split_count_host": (None, [vp, vp, c_uint32, c_uint32, vp, c_uint32, vp]),
    "mggcn_csr_block_split_fill_host": (None, [vp, vp, vp, c_uint32, c_uint32, vp, c_uint32, vp,
                                               ctypes.POINTER(vp), ctypes.POINTER(vp)]),
    "mggcn_init_uniform_host": (None, [vp, c_size_t, c_size_t, c_float]),
}

_lib = None


def load():
    """dlopen the engine and type every entry point.  torch is imported first so
    that a process that also uses torch shares ONE HIP runtime (torch bundles
    libamdhip64.so.7; the loader de-duplicates by soname)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise engine_error(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C mg-gcn_amd/csrc`).  There is no CPU fallback.")
    try:
        import torch  # noqa: F401  (one HIP runtime per process)
    except Exception:
        pass
    lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)      # AttributeError here == the ABI lost a symbol
        fn.restype = res
        fn.argtypes = args
    if lib.mggcn_abi_version() != 1:
        raise engine_error("libmggcn_hip.so ABI version mismatch")
    _lib = lib
    return lib


def require_gpu() -> int:
    """Number of visible GPUs; raises (loudly) when there is none."""
    n = load().mggcn_device_count()
    if n <= 0:
        raise engine_error("no MI355X visible to the HIP runtime: the MG-GCN engine has no CPU path")
    return n
