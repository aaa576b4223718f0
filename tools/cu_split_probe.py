"""Probe (NOT part of the product): the graph product and the highway gate's GEMMs on DISJOINT CU sets
(hipExtStreamCreateWithCUMask), TwitterUS shape, F = 300 -- the question of VERDICT r03 item 5: the SpMM is fabric-bound,
so is its time flat down to ~3/4 of the CUs, and do the gate's GEMMs (which never depend on the SpMM of their own layer:
reference gcnmodel.py:130 vs :285) then run for free on the rest?
    python tools/cu_split_probe.py [--sets low|spread] > profiles/r04_cu_split_raw.txt
Forward pair : tanh(A.Z + bh)  ||  T = sigmoid(H.Wt + bt)                       (gcnmodel.py:130-136 || :285-286)
Backward trio: A^T.dS          ||  dWt = H^T.dU ; dH += dU.Wt^T                 (their gradients)
CU sets: 'spread' takes the first n/8 bits of every 32-bit mask word, 'low' the first n bits (tools/micro/cu_mask_map.hip
prints which physical CUs either pattern selects)."""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from geographconv_amd import _ffi, ops, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--sets', default='spread')
ap.add_argument('--reps', type=int, default=20)
args = ap.parse_args()

dev = torch.device('cuda:0')
hip = C.CDLL('libamdhip64.so')
lib = _ffi.lib()
p = ops._p


def cu_sets(n_spmm):
    """-> (bits for the SpMM stream, bits for the GEMM stream), disjoint, together all 256."""
    if args.sets == 'low':
        a = list(range(n_spmm))
    else:
        per = [n_spmm // 8 + (1 if w < n_spmm % 8 else 0) for w in range(8)]
        a = [32 * w + i for w in range(8) for i in range(per[w])]
    sa = set(a)
    return a, [i for i in range(256) if i not in sa]


def masked_stream(cus):
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
N, F = s.N, 300
Z = ops.DMat.empty(N, F, dev, ld=320)
Z.t[:, :F].copy_(torch.from_numpy(rng.randn(N, F).astype(np.float32)))
S = ops.DMat(N, F, dev)
H = ops.DMat.from_numpy(np.tanh(rng.randn(N, F)).astype(np.float32), dev)
dU = ops.DMat.from_numpy((rng.randn(N, F) * 1e-3).astype(np.float32), dev)
dH = ops.DMat(N, F, dev)
Wt = ops.DMat.from_numpy((rng.randn(F, F) * 0.05).astype(np.float32), dev)
Wh = ops.DMat.from_numpy((rng.randn(F, F) * 0.05).astype(np.float32), dev)
bt = torch.full((F,), -4.0, device=dev)
bh = torch.zeros(F, device=dev)
T = ops.DMat(N, F, dev)
Z2 = ops.DMat.empty(N, F, dev, ld=320)
dW = ops.DMat(F, F, dev)
ws_sp = dA._ws.get(lib.geogcn_spmm_workspace_bytes(dA._plan, F))
ws_tn = torch.empty(lib.geogcn_gemm_workspace_bytes(1, 0, F, F, N, 0), dtype=torch.uint8, device=dev)
ws_dual = torch.empty(lib.geogcn_gemm_dual_workspace_bytes(0, N, F, F, F, 0), dtype=torch.uint8, device=dev)
torch.cuda.synchronize()


def spmm(st):
    rc = lib.geogcn_spmm_csr_f32(dA._plan, N, N, dA.nnz, p(dA.rowptr), p(dA.colidx), p(dA.val), p(Z.t), Z.ld, p(S.t), S.ld, F,
                                 p(bh), 1, p(ws_sp), ws_sp.numel(), st)
    assert rc == 0, rc


def gemm_gate(st):          # T = sigmoid(H.Wt + bt): the staged kernel (what a single 300-wide product runs on)
    rc = lib.geogcn_gemm_f32(0, 0, N, F, F, p(H.t), H.ld, p(Wt.t), Wt.ld, p(T.t), T.ld, p(bt), 2, 0, 0, None, 0, st)
    assert rc == 0, rc


def gemm_dual(st):          # (Z, T) in one launch on the whole-rows kernel: what the step runs today
    rc = lib.geogcn_gemm_dual_f32(0, N, F, F, F, p(H.t), H.ld, p(Wh.t), Wh.ld, p(Wt.t), Wt.ld, p(Z2.t), Z2.ld, p(T.t), T.ld,
                                  None, 0, p(bt), 2, 0, p(ws_dual), ws_dual.numel(), st)
    assert rc == 0, rc


def gemm_tn(st):            # dWt = H^T.dU
    rc = lib.geogcn_gemm_f32(1, 0, F, F, N, p(H.t), H.ld, p(dU.t), dU.ld, p(dW.t), dW.ld, None, 0, 0, 0, p(ws_tn), ws_tn.numel(), st)
    assert rc == 0, rc


def gemm_nt_acc(st):        # dH += dU.Wt^T
    rc = lib.geogcn_gemm_f32(0, 1, N, F, F, p(dU.t), dU.ld, p(Wt.t), Wt.ld, p(dH.t), dH.ld, None, 0, 1, 0, None, 0, st)
    assert rc == 0, rc


def timed(fn, reps=None):
    reps = reps or args.reps
    for _ in range(3):
        fn()
        sync()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        sync()
        ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts))


full = masked_stream(range(256))
full2 = masked_stream(range(256))
null = timed(lambda: None)
print("# CU sets: %s;  %d reps, median of launch -> hipDeviceSynchronize wall times; an empty sync costs %.3f ms (not subtracted)" % (args.sets, args.reps, null))
print("# one stream, all 256 CUs: spmm %.3f | gate NN %.3f | TN %.3f | NT+acc %.3f | dual NN (whole rows) %.3f ms" % (
    timed(lambda: spmm(full)), timed(lambda: gemm_gate(full)), timed(lambda: gemm_tn(full)), timed(lambda: gemm_nt_acc(full)),
    timed(lambda: gemm_dual(full))))
seq_f = timed(lambda: (spmm(full), gemm_gate(full)))
seq_b = timed(lambda: (spmm(full), gemm_tn(full), gemm_nt_acc(full)))
two_f = timed(lambda: (spmm(full), gemm_gate(full2)))
two_b = timed(lambda: (spmm(full), gemm_tn(full2), gemm_nt_acc(full2)))
print("# sequential on one stream: forward pair %.3f ms, backward trio %.3f ms" % (seq_f, seq_b))
print("# two unmasked streams    : forward pair %.3f ms, backward trio %.3f ms" % (two_f, two_b))
print("| CUs spmm / gemm | spmm alone | gate NN alone | fwd pair concurrent | vs sequential | TN + NT alone | bwd trio concurrent | vs sequential |")
print("|---|---|---|---|---|---|---|---|")
for n_sp in (224, 208, 192, 160, 128):
    a, b = cu_sets(n_sp)
    sa, sb = masked_stream(a), masked_stream(b)
    t_sp = timed(lambda: spmm(sa))
    t_g = timed(lambda: gemm_gate(sb))
    t_f = timed(lambda: (spmm(sa), gemm_gate(sb)))
    t_gb = timed(lambda: (gemm_tn(sb), gemm_nt_acc(sb)))
    t_b = timed(lambda: (spmm(sa), gemm_tn(sb), gemm_nt_acc(sb)))
    print("| %d / %d | %.3f | %.3f | %.3f | %+.3f | %.3f | %.3f | %+.3f |" % (n_sp, 256 - n_sp, t_sp, t_g, t_f, t_f - seq_f, t_gb, t_b, t_b - seq_b),
          flush=True)
# the GEMM on the LARGER set, the SpMM squeezed: does the fabric-bound kernel need its CUs?
for n_sp in (96, 64):
    a, b = cu_sets(n_sp)
    sa, sb = masked_stream(a), masked_stream(b)
    print("| %d / %d | %.3f | %.3f | %.3f | %+.3f | - | - | - |" % (n_sp, 256 - n_sp, timed(lambda: spmm(sa)), timed(lambda: gemm_gate(sb)),
                                                                 timed(lambda: (spmm(sa), gemm_gate(sb))), timed(lambda: (spmm(sa), gemm_gate(sb))) - seq_f), flush=True)
