"""GEMM-only probe at TwitterUS shape for rocprofv3 --pmc passes.  python tools/gemm_probe.py [nn|nt|tn] [N] [precision]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from geographconv_amd import ops  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else 'nn'
Fo = int(sys.argv[2]) if len(sys.argv) > 2 else 300
prec = sys.argv[3] if len(sys.argv) > 3 else 'f32'
N = 440000
dev = torch.device('cuda:0')
rng = np.random.RandomState(1)
H = ops.DMat.from_numpy(rng.randn(N, 300).astype(np.float32), dev)
W = ops.DMat.from_numpy((rng.randn(300, Fo) * 0.05).astype(np.float32), dev)
Z = ops.DMat.from_numpy(rng.randn(N, Fo).astype(np.float32), dev)
for _ in range(3):
    if mode == 'nn':
        ops.gemm(H, W, out=Z, precision=prec)
    elif mode == 'nt':
        ops.gemm(Z, W, out=H, transB=True, precision=prec)
    else:
        ops.gemm(H, Z, transA=True)
torch.cuda.synchronize()
