"""Probe (NOT part of the product): the bf16-configuration GEMMs at the configs[4] width (N x 600 x 600), timed and --
under `rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE|TCC_HIT_sum TCC_MISS_sum` -- with their fabric traffic per launch.
    python tools/bf16_gemm_probe.py [N] [F] [reps]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from geographconv_amd import ops  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 440000
F = int(sys.argv[2]) if len(sys.argv) > 2 else 600
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
LD = int(sys.argv[4]) if len(sys.argv) > 4 else None      # pitch of the streamed operands (default: roundup4(F))
dev = torch.device('cuda:0')
rng = np.random.RandomState(1)
H = ops.DMat.empty(N, F, dev, ld=LD); H.t.zero_(); H.t[:, :F].normal_()
G = ops.DMat.empty(N, F, dev, ld=LD); G.t.zero_(); G.t[:, :F].normal_()
W = ops.DMat.from_numpy((rng.randn(F, F) * 0.05).astype(np.float32), dev)
b = torch.zeros(F, device=dev)
Zb = ops.HMat(N, F, dev)
T = ops.DMat.empty(N, F, dev, ld=LD)
dW = ops.DMat.empty(F, F, dev)


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return a.elapsed_time(e) / reps


fl = 2.0 * N * F * F
cases = [
    ('NN  Z(bf16) = H.W', lambda: ops.gemm(H, W, out=Zb, precision='bf16'), 4 * N * F + 2 * N * F),
    ('NN  T = sigmoid(H.W + b)', lambda: ops.gemm(H, W, out=T, bias=b, act=ops.ACT_SIGMOID, precision='bf16'), 8 * N * F),
    ('NT  dH = G.W^T', lambda: ops.gemm(G, W, out=T, transB=True, precision='bf16'), 8 * N * F),
    ('TN  dW = H^T.G', lambda: ops.gemm(H, G, out=dW, transA=True, precision='bf16'), 8 * N * F),
]
for name, fn, alg in cases:
    ms = timed(fn)
    print('%-28s %.3f ms  %.0f TFLOP/s   algorithmic %.2f GB -> %.0f GB/s = %.1f %% of 8 TB/s'
          % (name, ms, fl / ms / 1e9, alg / 1e9, alg / ms / 1e6, alg / ms / 1e6 / 80), flush=True)
