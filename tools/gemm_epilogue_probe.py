"""What the epilogues of the fused split-bf16 launches cost at the TwitterUS size (round 6; measurement tool, not part of the product):
the dual H.[Wh | Wt] with / without the gate's bias + sigmoid, the k-concatenated dH with / without the carry and the tanh gradient."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from geographconv_amd import ops  # noqa: E402

dev = torch.device('cuda:0')
M, F = 440000, 300
rng = np.random.RandomState(0)


def t(fn, reps=9):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); ev.append((a, b))
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2]


H = ops.DMat.empty(M, F, dev); H.t.normal_(0, 0.3)
G = ops.DMat.empty(M, F, dev); G.t.normal_(0, 0.3)
T = ops.DMat.empty(M, F, dev); T.t.uniform_(0, 1)
Y = ops.DMat.empty(M, F, dev); Y.t.uniform_(-1, 1)
keep = (torch.rand(M, F, device=dev) < 0.5).to(torch.uint8)
W0 = ops.DMat.from_numpy((rng.randn(F, F) * 0.1).astype(np.float32), dev)
W1 = ops.DMat.from_numpy((rng.randn(F, F) * 0.1).astype(np.float32), dev)
bias = torch.zeros(ops.pad4(F), device=dev)
o0, o1 = ops.DMat.empty(M, F, dev, ld=ops.gather_ld(F)), ops.DMat.empty(M, F, dev)
out = ops.DMat.empty(M, F, dev)
p = 'bf16x3'
print('dual, no epilogue           %.3f' % t(lambda: ops.gemm_dual(H, W0, W1, out0=o0, out1=o1, precision=p)))
print('dual, bias + sigmoid        %.3f' % t(lambda: ops.gemm_dual(H, W0, W1, out0=o0, out1=o1, bias1=bias, act1=ops.ACT_SIGMOID, precision=p)))
print('dual, sigmoid only          %.3f' % t(lambda: ops.gemm_dual(H, W0, W1, out0=o0, out1=o1, act1=ops.ACT_SIGMOID, precision=p)))
print('kcat plain                  %.3f' % t(lambda: ops.gemm_kcat(H, W0, G, W1, out=out, transB=True, precision=p)))
print('kcat accumulate             %.3f' % t(lambda: ops.gemm_kcat(H, W0, G, W1, out=out, transB=True, accumulate=True, precision=p)))
carry = ops.GateCarry(G, T)
print('kcat + carry                %.3f' % t(lambda: ops.gemm_kcat(H, W0, G, W1, out=out, transB=True, gate_carry=carry, precision=p)))
print('kcat + carry + tanh bwd     %.3f' % t(lambda: ops.gemm_kcat(H, W0, G, W1, out=o0, transB=True, gate_carry=carry, tanh_bwd=(Y, keep, 2.0), precision=p)))
print('single A.B^T                %.3f' % t(lambda: ops.gemm(H, W0, out=out, transB=True, precision=p)))
print('single A.B^T + carry        %.3f' % t(lambda: ops.gemm(H, W0, out=out, transB=True, precision=p, gate_carry=carry)))
print('dual tn                     %.3f' % t(lambda: ops.gemm_dual(H, G, T, transA=True, precision=p)))
