"""Probe (NOT part of the product): how do the SpMM and the fp32 GEMM scale with the number of CUs they may use
(hipExtStreamCreateWithCUMask), and what does running them CONCURRENTLY on disjoint CU sets give?
    python tools/cu_mask_probe.py"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from geographconv_amd import _ffi, ops, synth  # noqa: E402

dev = torch.device('cuda:0')
hip = C.CDLL('libamdhip64.so')
lib = _ffi.lib()


def masked_stream(cus):
    """cus: iterable of CU indices (0..255) -> hipStream_t restricted to them."""
    words = (C.c_uint32 * 8)()
    for c in cus:
        words[c // 32] |= (1 << (c % 32))
    st = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), C.c_uint32(8), words)
    assert rc == 0, rc
    return st


def sync():
    assert hip.hipDeviceSynchronize() == 0


s = synth.SHAPES['twus']
A = synth.powerlaw_ahat(s.N, s.E_target)
dA = ops.CSR(A, dev)
rng = np.random.RandomState(1)
Z = ops.DMat.empty(s.N, 300, dev, ld=320)
Z.t[:, :300].copy_(torch.from_numpy(rng.randn(s.N, 300).astype(np.float32)))
S = ops.DMat(s.N, 300, dev)
H = ops.DMat.from_numpy(rng.randn(s.N, 300).astype(np.float32), dev)
W = ops.DMat.from_numpy((rng.randn(300, 300) * 0.05).astype(np.float32), dev)
U = ops.DMat(s.N, 300, dev)
ws = dA._ws.get(lib.geogcn_spmm_workspace_bytes(dA._plan, 300))
p = ops._p


def spmm(st):
    rc = lib.geogcn_spmm_csr_f32(dA._plan, s.N, s.N, dA.nnz, p(dA.rowptr), p(dA.colidx), p(dA.val), p(Z.t), Z.ld, p(S.t),
                                 S.ld, 300, None, 0, p(ws), ws.numel(), st)
    assert rc == 0


def gemm(st):
    rc = lib.geogcn_gemm_f32(0, 0, s.N, 300, 300, p(H.t), H.ld, p(W.t), W.ld, p(U.t), U.ld, None, 0, 0, 0, None, 0, st)
    assert rc == 0


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    sync()
    return (time.perf_counter() - t0) / reps * 1e3


torch.cuda.synchronize()
# CU numbering in the mask: bit i = CU i; XCD striping is not documented -- try both "first n" and "every k-th"
for n in (256, 192, 128, 64):
    first = masked_stream(range(n))
    step = 256 // n if 256 % n == 0 else None
    print("CUs %3d (first n):  spmm %.3f ms   gemm %.3f ms" % (n, timed(lambda: spmm(first)), timed(lambda: gemm(first))), flush=True)
    if step and step > 1:
        inter = masked_stream(range(0, 256, step))
        print("CUs %3d (every %d):  spmm %.3f ms   gemm %.3f ms" % (n, step, timed(lambda: spmm(inter)), timed(lambda: gemm(inter))), flush=True)
# concurrent: disjoint halves / thirds
for na in (128, 96, 64):
    sa = masked_stream(range(0, na))
    sb = masked_stream(range(na, 256))
    t_pair = timed(lambda: (spmm(sa), gemm(sb)))
    print("concurrent: spmm on CUs [0,%d) + gemm on [%d,256): %.3f ms per pair  (alone on those sets: spmm %.3f, gemm %.3f)" % (
        na, na, t_pair, timed(lambda: spmm(sa)), timed(lambda: gemm(sb))), flush=True)
# concurrent without masks (two plain streams)
s1, s2 = masked_stream(range(256)), masked_stream(range(256))
print("concurrent, no partition (two full-mask streams): %.3f ms per pair" % timed(lambda: (spmm(s1), gemm(s2))), flush=True)
print("sequential on one stream: %.3f ms per pair" % timed(lambda: (spmm(s1), gemm(s1))), flush=True)


# the backward pairing of a highway block: A^T.dS on one stream, the gate's two gradient GEMMs (TN + NT) on another
dW = ops.DMat(300, 300, dev)
wtn = torch.empty(lib.geogcn_gemm_workspace_bytes(1, 0, 300, 300, s.N, 0), dtype=torch.uint8, device=dev)


def gemm_tn(st):
    rc = lib.geogcn_gemm_f32(1, 0, 300, 300, s.N, p(H.t), H.ld, p(U.t), U.ld, p(dW.t), dW.ld, None, 0, 0, 0, p(wtn), wtn.numel(), st)
    assert rc == 0


def gemm_nt_acc(st):
    rc = lib.geogcn_gemm_f32(0, 1, s.N, 300, 300, p(U.t), U.ld, p(W.t), W.ld, p(H.t), H.ld, None, 0, 1, 0, None, 0, st)
    assert rc == 0


print("backward trio sequential (spmm, tn, nt+acc) on one stream: %.3f ms" % timed(lambda: (spmm(s1), gemm_tn(s1), gemm_nt_acc(s1))), flush=True)
print("backward trio, spmm on stream 1, (tn, nt+acc) on stream 2:  %.3f ms" % timed(lambda: (spmm(s1), gemm_tn(s2), gemm_nt_acc(s2))), flush=True)
print("backward trio, GEMMs launched first:                        %.3f ms" % timed(lambda: (gemm_tn(s2), gemm_nt_acc(s2), spmm(s1))), flush=True)
