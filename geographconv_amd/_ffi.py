"""ctypes binding of libgeogcn.so (include/geogcn.h).  No torch types cross this boundary: the
callers in ops.py pass ``tensor.data_ptr()`` integers and the raw hipStream_t of torch's current
stream.  There is NO fallback: if the library is missing or a call fails, this raises."""
from __future__ import annotations

import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libgeogcn.so')
HEADER_PATH = os.path.join(os.path.dirname(_HERE), 'include', 'geogcn.h')

ACT_NONE, ACT_TANH, ACT_SIGMOID, ACT_SELU, ACT_RELU = 0, 1, 2, 3, 4
GEMM_F32, GEMM_BF16X3, GEMM_BF16 = 0, 1, 2

c_i32, c_i64, c_f32, c_sz, c_ptr = C.c_int32, C.c_int64, C.c_float, C.c_size_t, C.c_void_p
c_u64 = C.c_uint64
COMM_ID_BYTES = 128        # GEOGCN_COMM_ID_BYTES

# name -> (restype, argtypes); mirrors include/geogcn.h one to one
SIGNATURES = {
    'geogcn_version': (c_i32, []),
    'geogcn_last_error': (C.c_char_p, []),
    'geogcn_spmm_plan_create': (c_i32, [c_i32, c_ptr, c_i32, c_i32, c_i32, C.POINTER(c_ptr)]),
    'geogcn_spmm_plan_destroy': (None, [c_ptr]),
    'geogcn_spmm_plan_num_long_rows': (c_i64, [c_ptr]),
    'geogcn_spmm_plan_num_chunks': (c_i64, [c_ptr]),
    'geogcn_spmm_workspace_bytes': (c_sz, [c_ptr, c_i32]),
    'geogcn_spmm_csr_f32': (c_i32, [c_ptr, c_i32, c_i32, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_i64,
                                    c_ptr, c_i64, c_i32, c_ptr, c_i32, c_ptr, c_sz, c_ptr]),
    'geogcn_spmm_csr_softmax_f32': (c_i32, [c_ptr, c_i32, c_i32, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_i64,
                                            c_ptr, c_i64, c_i32, c_ptr, c_ptr, c_ptr, c_sz, c_ptr]),
    'geogcn_spmm_csr_acc_f32': (c_i32, [c_ptr, c_i32, c_i32, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_i64,
                                        c_ptr, c_i64, c_i32, c_ptr, c_i32, c_ptr, c_sz, c_ptr]),
    'geogcn_spmm_hot_capacity': (c_i32, [c_i32]),
    'geogcn_spmm_csr_hot_f32': (c_i32, [c_i32, c_i32, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_i32, c_ptr, c_ptr, c_i64, c_i32,
                                        c_ptr, c_i32, c_ptr]),
    'geogcn_spmm_csr_hot_dropout_f32': (c_i32, [c_i32, c_i32, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_i32, c_ptr, c_ptr, c_ptr, c_i64,
                                                c_i32, c_ptr, c_i32, c_f32, c_ptr, c_ptr, c_u64, c_u64, c_ptr, c_i64, c_i64, c_ptr]),
    'geogcn_xt_plan_create': (c_i32, [c_i32, c_i32, c_ptr, c_ptr, c_i32, C.POINTER(c_ptr)]),
    'geogcn_xt_plan_destroy': (None, [c_ptr]),
    'geogcn_xt_workspace_bytes': (c_sz, [c_ptr]),
    'geogcn_xt_dot_f32': (c_i32, [c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_sz, c_ptr]),
    'geogcn_spmm_csr_bf16b': (c_i32, [c_ptr, c_i32, c_i32, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_i64,
                                      c_ptr, c_i64, c_i32, c_ptr, c_i32, c_ptr, c_sz, c_ptr]),
    'geogcn_cast_bf16_f32': (c_i32, [c_i64, c_i32, c_ptr, c_i64, c_ptr, c_i64, c_ptr]),
    'geogcn_spmm_csr_highway_f32': (c_i32, [c_ptr, c_i32, c_i32, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i32, c_i32,
                                            c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_sz, c_ptr]),
    'geogcn_timer_create': (c_i32, [c_i32, C.POINTER(c_ptr)]),
    'geogcn_timer_destroy': (None, [c_ptr]),
    'geogcn_spmm_plan_attach_timer': (c_i32, [c_ptr, c_ptr, c_i32]),
    'geogcn_timer_read_ms': (c_i32, [c_ptr, c_ptr, c_i32, C.POINTER(c_i32)]),
    'geogcn_gemm_workspace_bytes': (c_sz, [c_i32, c_i32, c_i64, c_i64, c_i64, c_i32]),
    'geogcn_gemm_f32': (c_i32, [c_i32, c_i32, c_i64, c_i64, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_ptr,
                                c_i64, c_ptr, c_i32, c_i32, c_i32, c_ptr, c_sz, c_ptr]),
    'geogcn_gemm_f32_bf16c': (c_i32, [c_i32, c_i64, c_i64, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_ptr,
                                      c_i32, c_ptr, c_sz, c_ptr]),
    'geogcn_gemm_panels_f32': (c_i32, [c_i32, c_i64, c_i64, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_i32, c_i32, c_i32,
                                       c_ptr, c_i32, c_i32, c_ptr, c_sz, c_ptr]),
    'geogcn_gemm_dual_workspace_bytes': (c_sz, [c_i32, c_i64, c_i64, c_i64, c_i64, c_i32]),
    'geogcn_gemm_dual_f32': (c_i32, [c_i32, c_i64, c_i64, c_i64, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64,
                                     c_ptr, c_i64, c_ptr, c_i32, c_ptr, c_i32, c_i32, c_ptr, c_sz, c_ptr]),
    'geogcn_gemm_kcat_workspace_bytes': (c_sz, [c_i32, c_i64, c_i64, c_i64, c_i64, c_i32]),
    'geogcn_gemm_kcat_f32': (c_i32, [c_i32, c_i64, c_i64, c_i64, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64,
                                     c_ptr, c_i64, c_i32, c_i32, c_ptr, c_sz, c_ptr]),
    'geogcn_gemm_kcat_gated_f32': (c_i32, [c_i32, c_i64, c_i64, c_i64, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64,
                                           c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_i32, c_ptr, c_sz, c_ptr]),
    'geogcn_gemm_dual_bf16_workspace_bytes': (c_sz, [c_i64, c_i64, c_i64, c_i64]),
    'geogcn_gemm_dual_bf16': (c_i32, [c_i64, c_i64, c_i64, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_i32,
                                      c_ptr, c_i64, c_ptr, c_i32, c_ptr, c_sz, c_ptr]),
    'geogcn_gemm_gated_f32': (c_i32, [c_i32, c_i64, c_i64, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64,
                                      c_i32, c_ptr, c_sz, c_ptr]),
    'geogcn_gemm_kcat_gated_tanhbwd_f32': (c_i32, [c_i32, c_i64, c_i64, c_i64, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64,
                                                   c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_f32, c_i32, c_ptr,
                                                   c_sz, c_ptr]),
    'geogcn_gate_carry_f32': (c_i32, [c_i64, c_i32, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_ptr]),
    'geogcn_bias_act_f32': (c_i32, [c_i64, c_i32, c_ptr, c_i64, c_ptr, c_i32, c_ptr, c_i64, c_ptr]),
    'geogcn_highway_fwd_f32': (c_i32, [c_i64, c_i32, c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_ptr]),
    'geogcn_highway_bwd_f32': (c_i32, [c_i64, c_i32, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_i64, c_ptr,
                                       c_ptr, c_ptr, c_ptr, c_ptr, c_sz, c_ptr]),
    'geogcn_highway_bwd_bf16s_f32': (c_i32, [c_i64, c_i32, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_i64, c_ptr,
                                             c_ptr, c_ptr, c_ptr, c_ptr, c_sz, c_ptr]),
    'geogcn_highway_bwd_workspace_bytes': (c_sz, [c_i64, c_i32]),
    'geogcn_act_bwd_f32': (c_i32, [c_i64, c_i32, c_ptr, c_ptr, c_i64, c_i32, c_ptr, c_f32, c_ptr, c_i64, c_ptr]),
    'geogcn_act_bwd_colsum_f32': (c_i32, [c_i64, c_i32, c_ptr, c_ptr, c_i64, c_i32, c_ptr, c_f32, c_ptr, c_i64, c_ptr,
                                          c_ptr, c_sz, c_ptr]),
    'geogcn_add_inplace_f32': (c_i32, [c_i64, c_ptr, c_ptr, c_ptr]),
    'geogcn_colsum_workspace_bytes': (c_sz, [c_i64, c_i32]),
    'geogcn_colsum_rowblocks_f32': (c_i32, [c_i64, c_i32, c_ptr, c_i64, c_ptr, c_ptr, c_sz, c_ptr]),
    'geogcn_colsum_f32': (c_i32, [c_i64, c_i32, c_ptr, c_i64, c_ptr, c_ptr, c_sz, c_ptr]),
    'geogcn_dropout_mask_philox': (c_i32, [c_i64, c_i32, c_f32, c_u64, c_u64, c_ptr, c_ptr]),
    'geogcn_dropout_mask_philox_ctr': (c_i32, [c_i64, c_i32, c_f32, c_u64, c_ptr, c_i64, c_i64, c_ptr, c_ptr]),
    'geogcn_counter_add_i64': (c_i32, [c_ptr, c_i64, c_ptr]),
    'geogcn_dropout_csr_f32': (c_i32, [c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_i32, c_f32, c_u64, c_u64, c_ptr]),
    'geogcn_dropout_panel_f32': (c_i32, [c_i64, c_i32, c_ptr, c_i64, c_ptr, c_i64, c_f32, c_u64, c_u64, c_ptr, c_ptr]),
    'geogcn_dropout_apply_f32': (c_i32, [c_i64, c_i32, c_ptr, c_i64, c_ptr, c_f32, c_ptr, c_ptr]),
    'geogcn_softmax_rows_f32': (c_i32, [c_i64, c_i32, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_ptr]),
    'geogcn_ce_metrics_workspace_bytes': (c_sz, [c_i64]),
    'geogcn_ce_metrics_f32': (c_i32, [c_i32, c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_ptr,
                                      c_sz, c_ptr]),
    'geogcn_softmax_ce_bwd_f32': (c_i32, [c_i64, c_i32, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_f32, c_ptr,
                                          c_i64, c_ptr]),
    'geogcn_softmax_ce_bwd_db_workspace_bytes': (c_sz, [c_i32]),
    'geogcn_softmax_ce_bwd_db_f32': (c_i32, [c_i64, c_i32, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_f32, c_ptr, c_i64, c_ptr,
                                             c_ptr, c_sz, c_ptr]),
    'geogcn_softmax_ce_rows_bwd_db_f32': (c_i32, [c_i32, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_f32, c_ptr, c_i64, c_ptr, c_ptr, c_sz,
                                                  c_ptr]),
    'geogcn_gather_rows_f32': (c_i32, [c_i32, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_ptr]),
    'geogcn_scatter_rows_f32': (c_i32, [c_i32, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_ptr]),
    'geogcn_comm_available': (c_i32, []),
    'geogcn_comm_unique_id': (c_i32, [c_ptr, c_sz]),
    'geogcn_comm_init_rank': (c_i32, [c_ptr, c_i32, c_i32, C.POINTER(c_ptr)]),
    'geogcn_comm_destroy': (None, [c_ptr]),
    'geogcn_comm_world': (c_i32, [c_ptr]),
    'geogcn_comm_rank': (c_i32, [c_ptr]),
    'geogcn_comm_allreduce_sum_f32': (c_i32, [c_ptr, c_ptr, c_i64, c_ptr]),
    'geogcn_comm_allgather': (c_i32, [c_ptr, c_ptr, c_ptr, c_i64, c_ptr]),
    'geogcn_comm_alltoall': (c_i32, [c_ptr, c_ptr, c_ptr, c_i64, c_ptr]),
    'geogcn_comm_alltoallv': (c_i32, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    'geogcn_pack_panels_f32': (c_i32, [c_i64, c_i64, c_i32, c_ptr, c_i64, c_i32, c_i32, c_ptr, c_ptr]),
    'geogcn_unpack_panels_f32': (c_i32, [c_i64, c_i64, c_i32, c_ptr, c_i32, c_i32, c_ptr, c_i64, c_ptr]),
    'geogcn_adam_step_f32': (c_i32, [c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_f32, c_f32, c_f32, c_f32,
                                     c_i32, c_f32, c_f32, c_ptr]),
    'geogcn_adam_step_ctr_f32': (c_i32, [c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_f32, c_f32, c_f32, c_f32,
                                         c_ptr, c_f32, c_f32, c_ptr]),
    'geogcn_reg_penalty_f32': (c_i32, [c_i64, c_ptr, c_ptr, c_f32, c_f32, c_ptr, c_ptr, c_sz, c_ptr]),
}


ABI_VERSION = 3          # == GEOGCN_ABI_VERSION of include/geogcn.h (tests/test_abi.py holds the two together)


def header_symbols():
    """Every function name include/geogcn.h declares."""
    with open(HEADER_PATH) as f:
        text = f.read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(geogcn_[a-z0-9_]+)\s*\(', text)))


class GeoGcnError(RuntimeError):
    pass


_lib = None


def lib():
    """Load libgeogcn.so (once).  torch is imported first so that the library resolves
    libamdhip64.so.7 to the SAME HIP runtime torch already loaded (one runtime per process)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GeoGcnError(
            "libgeogcn.so not found at %s -- build it with `python -m geographconv_amd.build` "
            "(hipcc, gfx950). There is no CPU fallback for the GCN hot path." % LIB_PATH)
    import torch  # noqa: F401  (loads the HIP runtime)
    handle = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(handle, name)
        fn.restype = res
        fn.argtypes = args
    if handle.geogcn_version() != ABI_VERSION:
        raise GeoGcnError("libgeogcn.so ABI version %d != %d (a stale library: python -m geographconv_amd.build --force)"
                          % (handle.geogcn_version(), ABI_VERSION))
    _lib = handle
    return _lib


def check(rc: int, what: str = ''):
    if rc != 0:
        msg = lib().geogcn_last_error().decode('utf-8', 'replace')
        kind = 'argument error' if rc < 0 else ('hipError' if rc < 1000 else 'ncclResult + 1000')
        raise GeoGcnError("%s failed (%s %d): %s" % (what or 'geogcn call', kind, rc, msg))
