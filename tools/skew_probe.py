"""Does the relative alignment of the seven streams of highway_bwd matter?  (NOT part of the product.)  The same kernel on
buffers as the allocator hands them out (all 2 MB aligned) and on row views that start k rows into a larger allocation."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from geographconv_amd import ops  # noqa: E402

dev = torch.device('cuda:0')
n, F = 440000, 300


def mat(skew, ld=None):
    m = ops.DMat.empty(n + 64, F, dev, ld=ld)
    m.t.normal_()
    return m.rows(skew, skew + n)


def timed(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)
    return t[len(t) // 2], t[0], t[-1]


bS = torch.zeros(ops.pad4(F), device=dev)
bU = torch.zeros(ops.pad4(F), device=dev)
for name, skews in [('all aligned (skew 0)', [0] * 7), ('skews 0,3,7,13,19,29,37 rows', [0, 3, 7, 13, 19, 29, 37]),
                    ('skews 0,1,2,3,4,5,6 rows', [0, 1, 2, 3, 4, 5, 6]), ('skews 0,9,18,27,36,45,54', [0, 9, 18, 27, 36, 45, 54])]:
    G, T, Hc, H, dU, dH = (mat(k) for k in skews[:6])
    dS = mat(skews[6], ld=ops.gather_ld(F))
    ptrs = [hex(m.t.data_ptr() % (1 << 21)) for m in (G, T, Hc, H, dS, dU, dH)]
    med, lo, hi = timed(lambda: ops.highway_bwd(G, T, Hc, H, dS=dS, dU=dU, dHcarry=dH, dbS=bS, dbU=bU))
    print('%-34s %.3f ms (min %.3f max %.3f)  offsets in 2 MB: %s' % (name, med, lo, hi, ' '.join(ptrs)), flush=True)
# fresh allocations a few times: does the placement alone move it?
for trial in range(4):
    keep = [torch.empty(int(np.random.RandomState(trial).randint(1, 200)) * 1024 * 1024, device=dev, dtype=torch.uint8)]
    G, T, Hc, H, dU, dH = (mat(0) for _ in range(6))
    dS = mat(0, ld=ops.gather_ld(F))
    med, lo, hi = timed(lambda: ops.highway_bwd(G, T, Hc, H, dS=dS, dU=dU, dHcarry=dH, dbS=bS, dbU=bU))
    print('fresh allocation %d                 %.3f ms (min %.3f max %.3f)' % (trial, med, lo, hi), flush=True)
