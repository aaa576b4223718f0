"""Yardstick (NOT part of the product): libgeogcn's fp32 GEMMs at TwitterUS shape next to the vendor
library torch.mm dispatches to (rocBLAS / hipBLASLt, fp32, TF32 off).  Prints ms and TFLOP/s.
    python tools/gemm_yardstick.py [N] [F]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from geographconv_amd import ops  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 440000
F = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dev = torch.device('cuda:0')
torch.backends.cuda.matmul.allow_tf32 = False
rng = np.random.RandomState(1)
H = ops.DMat.from_numpy(rng.randn(N, F).astype(np.float32), dev)
W = ops.DMat.from_numpy((rng.randn(F, F) * 0.05).astype(np.float32), dev)
Z = ops.DMat.from_numpy(rng.randn(N, F).astype(np.float32), dev)
dW = ops.DMat.empty(F, F, dev)
# vendor operands: dense, unpadded
Ht = H.t[:, :F].contiguous()
Wt = W.t[:, :F].contiguous()
Zt = Z.t[:, :F].contiguous()


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


flops = 2.0 * N * F * F
out_v = torch.empty_like(Zt)
out_w = torch.empty(F, F, device=dev)
cases = [
    ('NN  Z = H.W', lambda: ops.gemm(H, W, out=Z), lambda: torch.mm(Ht, Wt, out=out_v)),
    ('NT  dH = dZ.W^T', lambda: ops.gemm(Z, W, out=H, transB=True), lambda: torch.mm(Zt, Wt.t(), out=out_v)),
    ('TN  dW = H^T.dZ', lambda: ops.gemm(H, Z, out=dW, transA=True), lambda: torch.mm(Ht.t(), Zt, out=out_w)),
]
print("N=%d F=%d  (%.1f GFLOP per product)" % (N, F, flops / 1e9))
for name, ours, vendor in cases:
    t1, t2 = timed(ours), timed(vendor)
    print("%-18s libgeogcn %.3f ms (%.1f TF)   torch.mm %.3f ms (%.1f TF)" % (name, t1, flops / t1 / 1e9, t2, flops / t2 / 1e9))
