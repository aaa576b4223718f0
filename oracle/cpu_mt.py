"""ctypes loader of oracle/cpu_mt.c -- the multi-threaded CPU leg of bench.py (`cpu_baseline_mt`).  TEST / BENCH
INFRASTRUCTURE ONLY: nothing under geographconv_amd/ imports it.  `patched()` swaps the oracle's two sparse products for
the OpenMP ones for the duration of a timed step (transposes are prepared once, outside the timed region, as a
multi-threaded implementation would keep them)."""
import contextlib
import ctypes as C
import os
import subprocess

import numpy as np
import scipy.sparse as sps

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, 'cpu_mt.c')
LIB = os.path.join(HERE, '_build', 'libcpu_mt.so')
_lib = None


def build(force=False):
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        subprocess.run(['gcc', '-O3', '-march=native', '-fopenmp', '-shared', '-fPIC', SRC, '-o', LIB], check=True)
    return LIB


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        _lib = C.CDLL(LIB)
        _lib.geogcn_cpu_threads.restype = C.c_int
        _lib.geogcn_cpu_spmm_f32.restype = None
        _lib.geogcn_cpu_spmm_f32.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                             C.c_int64, C.c_int64]
    return _lib


def threads():
    return int(lib().geogcn_cpu_threads())


def spmm(A: sps.csr_matrix, B: np.ndarray) -> np.ndarray:
    """A_csr . B on all host cores (float32)."""
    A = sps.csr_matrix(A)
    B = np.ascontiguousarray(B, dtype=np.float32)
    out = np.empty((A.shape[0], B.shape[1]), dtype=np.float32)
    indptr = np.ascontiguousarray(A.indptr, dtype=np.int32)
    indices = np.ascontiguousarray(A.indices, dtype=np.int32)
    data = np.ascontiguousarray(A.data, dtype=np.float32)
    lib().geogcn_cpu_spmm_f32(A.shape[0], indptr.ctypes.data, indices.ctypes.data, data.ctypes.data, B.ctypes.data, B.shape[1],
                              out.ctypes.data, out.shape[1], B.shape[1])
    return out


@contextlib.contextmanager
def patched(oracle_module, transposes):
    """Inside the block, oracle.spmm / oracle.spmm_t run on all cores.  `transposes`: {id(matrix): CSR of its transpose}
    prepared by the caller (outside the timed region)."""
    old = oracle_module.spmm, oracle_module.spmm_t
    oracle_module.spmm = lambda A, B: spmm(A, B) if B.dtype == np.float32 else old[0](A, B)
    oracle_module.spmm_t = lambda A, G: spmm(transposes[id(A)], G) if G.dtype == np.float32 else old[1](A, G)
    try:
        yield
    finally:
        oracle_module.spmm, oracle_module.spmm_t = old
