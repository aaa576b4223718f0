"""ctypes loader of oracle/cpu_mt.c -- the multi-threaded CPU leg of bench.py (`cpu_baseline_mt`).  TEST / BENCH
INFRASTRUCTURE ONLY: nothing under geographconv_amd/ imports it.

Two uses:
  * `patched()` swaps the oracle's two sparse products for the OpenMP ones (transposes prepared once, outside the timed
    region, as a multi-threaded implementation would keep them);
  * `f_train()` is one full training step of the 3x300-style HIGHWAY network with EVERY pass on all cores: the sparse
    products and the fused elementwise / softmax / column-sum passes through cpu_mt.c (OpenMP), the dense products through
    BLAS sgemm (NumPy), Adam through NumPy (3.3 M parameters).  Same mathematics, same association order as
    oracle.f_train (gcnmodel.py:353-411 and what autodiff derives for it); tests/test_oracle.py compares the two."""
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
_P = C.c_void_p
_I = C.c_int64


def build(force=False):
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        subprocess.run(['gcc', '-O3', '-march=native', '-fopenmp', '-shared', '-fPIC', SRC, '-o', LIB, '-lm'], check=True)
    return LIB


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
            build()
        _lib = C.CDLL(LIB)
        _lib.geogcn_cpu_threads.restype = C.c_int
        sig = {
            'geogcn_cpu_spmm_f32': [_I, _P, _P, _P, _P, _I, _P, _I, _I],
            'geogcn_cpu_bias_tanh_f32': [_I, _I, _P, _P, _P, _P, _P],
            'geogcn_cpu_highway_fwd_f32': [_I, _I, _P, _P, _P, _P, _P, _P, _P, _P],
            'geogcn_cpu_bias_softmax_f32': [_I, _I, _P, _P, _P, _P],
            'geogcn_cpu_ce_grad_f32': [_I, _I, _P, _P, _P, _I, C.c_float, _P],
            'geogcn_cpu_colsum_f32': [_I, _I, _P, _P],
            'geogcn_cpu_highway_bwd_f32': [_I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P],
            'geogcn_cpu_tanh_bwd_f32': [_I, _I, _P, _P, _P, _P, _P],
            'geogcn_cpu_add_f32': [_I, _P, _P, _P],
        }
        for name, args in sig.items():
            fn = getattr(_lib, name)
            fn.restype, fn.argtypes = None, args
    return _lib


def threads():
    return int(lib().geogcn_cpu_threads())


def _d(a):
    return None if a is None else a.ctypes.data


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def spmm(A: sps.csr_matrix, B: np.ndarray) -> np.ndarray:
    """A_csr . B on all host cores (float32)."""
    A = sps.csr_matrix(A)
    B = _f32(B)
    out = np.empty((A.shape[0], B.shape[1]), dtype=np.float32)
    indptr = np.ascontiguousarray(A.indptr, dtype=np.int32)
    indices = np.ascontiguousarray(A.indices, dtype=np.int32)
    data = _f32(A.data)
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


# --------------------------------------------------------------------------------------------
# one whole training step on all cores
# --------------------------------------------------------------------------------------------
def f_train(O, params, st, X, Xt, y_train, y_dev, A, At, train_idx, dev_idx, hid, p_drop=0.0, mask=None):
    """oracle.f_train (gcnmodel.py:409-410) for a HIGHWAY network without regularisation, every pass multi-threaded.
    `O` = the oracle module (parameter bookkeeping, metrics and Adam are shared with it), `Xt` / `At` = CSR of the transposes
    (prepared once by the caller).  -> (new_params, [loss_tr, acc_tr, loss_dev, acc_dev, P], grads) like oracle.f_train."""
    L = lib()
    params = [_f32(p) for p in params]
    (W0, b0), blocks, (Wo, bo) = O.split_params(params, hid, True)
    N = X.shape[0]
    F = hid[0]
    e = lambda *shape: np.empty(shape, dtype=np.float32)
    # ---- forward (oracle.forward) ----
    S0 = spmm(X, W0)                                                   # gcnmodel.py:39
    H0 = e(N, F)
    scale = None
    Hd = H0
    if p_drop > 0:
        scale = _f32(np.asarray(mask, dtype=np.float32) / np.float32(1.0 - p_drop))
        Hd = e(N, F)
    L.geogcn_cpu_bias_tanh_f32(N, F, _d(S0), _d(b0), _d(H0), _d(scale), _d(Hd) if scale is not None else None)      # :41-42, :357
    H = Hd
    tape = []
    for (Wh, bh, Wt, bt) in blocks:
        Z = H @ Wh                                                     # :126
        S = spmm(A, Z)                                                 # :130
        U = H @ Wt                                                     # :285
        Hc, T, Hout = e(N, F), e(N, F), e(N, F)
        L.geogcn_cpu_highway_fwd_f32(N, F, _d(S), _d(bh), _d(U), _d(bt), _d(H), _d(Hc), _d(T), _d(Hout))           # :136, :286, :266
        tape.append((H, Hc, T))
        H = Hout
    Zo = H @ Wo                                                        # :149
    So = spmm(A, Zo)                                                   # :153
    Cn = Wo.shape[1]
    logits, P = e(N, Cn), e(N, Cn)
    L.geogcn_cpu_bias_softmax_f32(N, Cn, _d(So), _d(bo), _d(logits), _d(P))                                        # :155-157
    l_tr, a_tr, _ = O.metrics(P, train_idx, y_train)                                                               # :376-382
    l_dev, a_dev, _ = O.metrics(P, dev_idx, y_dev)                                                                 # :378-381,389
    # ---- backward (oracle.backward) ----
    idx64 = np.ascontiguousarray(train_idx, dtype=np.int64)
    y64 = np.ascontiguousarray(y_train, dtype=np.int64)
    dSo = e(N, Cn)
    L.geogcn_cpu_ce_grad_f32(N, Cn, _d(P), _d(idx64), _d(y64), len(idx64), np.float32(1.0 / max(1, len(idx64))), _d(dSo))
    dbo = e(Cn)
    L.geogcn_cpu_colsum_f32(N, Cn, _d(dSo), _d(dbo))
    dZo = spmm(At, dSo)
    dWo = H.T @ dZo
    G = dZo @ Wo.T
    gb = []
    for (Wh, bh, Wt, bt), (Hin, Hc, T) in zip(reversed(blocks), reversed(tape)):
        dS, dU, dH, dbh, dbt = e(N, F), e(N, F), e(N, F), e(F), e(F)
        L.geogcn_cpu_highway_bwd_f32(N, F, _d(G), _d(T), _d(Hc), _d(Hin), _d(dS), _d(dU), _d(dH), _d(dbh), _d(dbt))
        dZ = spmm(At, dS)
        dWh = Hin.T @ dZ
        dWt = Hin.T @ dU
        t1 = dZ @ Wh.T
        L.geogcn_cpu_add_f32(N * F, _d(dH), _d(t1), _d(dH))            # dH + dZ.Wh^T ...
        t1 = dU @ Wt.T
        L.geogcn_cpu_add_f32(N * F, _d(dH), _d(t1), _d(dH))            # ... + dU.Wt^T   (oracle.backward's order)
        gb.append([dWt, dbt, dWh, dbh])
        G = dH
    dS0, db0 = e(N, F), e(F)
    L.geogcn_cpu_tanh_bwd_f32(N, F, _d(G), _d(scale), _d(H0), _d(dS0), _d(db0))
    dW0 = spmm(Xt, dS0)
    grads = [dW0, db0]
    for g in reversed(gb):
        grads += g
    grads += [dWo, dbo]
    new_params = O.adam_step(params, grads, st)
    return new_params, [l_tr, a_tr, l_dev, a_dev, P], grads
