"""SpMM-only probe at TwitterUS shape: used (a) under rocprofv3 --pmc to read HBM traffic of the
SpMM kernels, (b) to sweep the feature width F (feature-slab hypothesis).
   python tools/spmm_probe.py [--F 300] [--reps 5] [--sweep]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from geographconv_amd import ops, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--F', type=int, nargs='+', default=[300])
    ap.add_argument('--reps', type=int, default=5)
    ap.add_argument('--shape', default='twus')
    ap.add_argument('--long', type=int, default=256)
    ap.add_argument('--chunk', type=int, default=128)
    ap.add_argument('--aligned', type=int, default=1, help='use the line-aligned gather pitch for B')
    ap.add_argument('--bf16', type=int, default=0, help='gather a bf16 copy of B (geogcn_spmm_csr_bf16b)')
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    s = synth.SHAPES[args.shape]
    A = synth.powerlaw_ahat(s.N, s.E_target)
    dA = ops.CSR(A, dev, args.long, args.chunk)
    print('long rows', dA.n_long_rows, flush=True)
    rng = np.random.RandomState(1)
    for F in args.F:
        H = ops.DMat.empty(s.N, F, dev, ld=ops.gather_ld(F) if args.aligned else None)
        H.t[:, :F].copy_(torch.from_numpy(rng.randn(s.N, F).astype(np.float32)))
        out = ops.DMat(s.N, F, dev)
        if args.bf16:
            H = ops.cast_bf16(H)
        for _ in range(2):
            ops.spmm(dA, H, out=out)
        torch.cuda.synchronize()
        evs = []
        for _ in range(args.reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            ops.spmm(dA, H, out=out)
            b.record()
            evs.append((a, b))
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) for a, b in evs)
        med = ts[len(ts) // 2]
        alg = 8 * A.nnz + 4 * (s.N + 1) + 8 * s.N * F
        print('F=%4d  %.3f ms  (%.3f us/col)  alg %.0f GB/s = %.1f%% of 8TB/s; gather %.1f GB -> %.2f TB/s' %
              (F, med, med * 1e3 / F, alg / med / 1e6, alg / med / 1e6 / 80, A.nnz * F * 4 / 1e9,
               A.nnz * F * 4 / med / 1e9), flush=True)


if __name__ == '__main__':
    main()
