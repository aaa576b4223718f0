"""Why do the fused GEMMs run ~10 % slower inside a training step than in a loop of their own?  (NOT part of the product.)
The highway block's dual launch (H.[Wh|Wt], TwitterUS shape) timed with an event pair around EACH call:
  (a) back to back on one buffer set;  (b) each call after a graph product (fabric-bound gather);  (c) each call after a
  streaming elementwise kernel;  (d) back to back over ROTATING buffer sets (cold MALL / TLB reach);  (e) the data the model
  sees: H = tanh outputs (|x| <= 1, dense mantissas) against zeros and against randn.
    python tools/gemm_in_context.py [N] [F]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from geographconv_amd import ops, synth  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 440000
F = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dev = torch.device('cuda:0')
rng = np.random.RandomState(1)
SETS = 6


def mat(scale=1.0, kind='randn'):
    m = ops.DMat.empty(N, F, dev, ld=ops.gather_ld(F))
    if kind == 'zeros':
        m.t.zero_()
    elif kind == 'tanh':
        m.t.normal_().tanh_()
    else:
        m.t.normal_().mul_(scale)
    return m


Hs = [mat() for _ in range(SETS)]
Zs = [mat() for _ in range(SETS)]
Ts = [mat() for _ in range(SETS)]
W = ops.DMat.from_numpy((rng.randn(F, F) * 0.05).astype(np.float32), dev)
W2 = ops.DMat.from_numpy((rng.randn(F, F) * 0.05).astype(np.float32), dev)
bt = torch.full((ops.pad4(F),), -4.0, device=dev)
A = ops.CSR(synth.powerlaw_ahat(N, 10000000 if N == 440000 else N * 14), dev)
S = mat()
flops = 4.0 * N * F * F


def dual(i):
    ops.gemm_dual(Hs[i], W, W2, out0=Zs[i], out1=Ts[i], bias1=bt, act1=ops.ACT_SIGMOID)


def kcat(i):
    ops.gemm_kcat(Zs[i], W, Ts[i], W2, out=Hs[i], transB=True, accumulate=True)


def tn(i):
    ops.gemm_dual(Hs[i], Zs[i], Ts[i], out0=dW, out1=dW2, transA=True)


dW, dW2 = ops.DMat.empty(F, F, dev), ops.DMat.empty(F, F, dev)


def per_call(fn, before=None, rotate=False, reps=24):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for k in range(reps + 4):
        i = (k % SETS) if rotate else 0
        if before is not None:
            before(i)
        if k >= 4:
            ev[k - 4][0].record()
        fn(i)
        if k >= 4:
            ev[k - 4][1].record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)
    return t[len(t) // 2], t[0], t[-1]


for _ in range(100):
    dual(0)
torch.cuda.synchronize()
big = torch.empty(N * 320 * 2, device=dev)
contexts = [
    ('back to back, one buffer set', None, False),
    ('after a graph product (A.Z, F = 300)', lambda i: ops.spmm(A, Zs[i], out=S), False),
    ('after a streaming kernel (2.2 GB)', lambda i: big.mul_(1.0001), False),
    ('back to back, %d rotating buffer sets' % SETS, None, True),
    ('after a graph product, rotating sets', lambda i: ops.spmm(A, Zs[i], out=S), True),
]
for name, fn in [('dual NN  H.[Wh|Wt]', dual), ('kcat NT  += [dZ|dU].[Wh|Wt]^T', kcat), ('dual TN  H^T.[dZ|dU]', tn)]:
    for Hm in Hs:
        Hm.t.normal_()
    for cname, before, rot in contexts:
        med, lo, hi = per_call(fn, before, rot)
        print("%-32s %-42s %.3f ms (%.1f TF)  min %.3f max %.3f" % (name, cname, med, flops / med / 1e9, lo, hi), flush=True)

print("operand values (dual NN, back to back):")
for kind in ('zeros', 'randn', 'tanh'):
    Hs[0] = mat(kind=kind)
    med, lo, hi = per_call(dual)
    print("  H = %-6s %.3f ms (%.1f TF)" % (kind, med, flops / med / 1e9), flush=True)
