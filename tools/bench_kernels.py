"""Micro-benchmarks of the individual kernels at TwitterUS / CMU shape (run on the MI355X).
   python tools/bench_kernels.py [--shape twus|cmu] [--reps 20]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from geographconv_amd import ops, synth  # noqa: E402


def timeit(fn, reps, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2], ts[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--shape', default='twus')
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--long', type=int, default=256)
    ap.add_argument('--chunk', type=int, default=128)
    ap.add_argument('--precision', default='f32')
    ap.add_argument('--only', default='all', choices=['all', 'gemm', 'spmm', 'stream'])
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    s = synth.SHAPES[args.shape]
    t0 = time.time()
    A = synth.powerlaw_ahat(s.N, s.E_target)
    X = synth.bow_x(s.N, s.V, s.mean_nnz)
    print('generated in %.1fs' % (time.time() - t0), flush=True)
    N, E = s.N, A.nnz
    res = {'shape': args.shape, 'N': N, 'nnz_A': int(E), 'nnz_X': int(X.nnz)}
    do = lambda k: args.only in ('all', k)
    dA = ops.CSR(A, dev, args.long, args.chunk)
    print('long rows', dA.n_long_rows, 'chunks', dA.n_chunks, flush=True)
    rng = np.random.RandomState(1)
    for F in ((300, s.C) if do('spmm') else ()):
        H = ops.DMat.from_numpy(rng.randn(N, F).astype(np.float32), dev)
        out = ops.DMat(N, F, dev)
        med, mn = timeit(lambda: ops.spmm(dA, H, out=out), args.reps)
        bytes_alg = 8 * E + 4 * (N + 1) + 4 * N * F + 4 * N * F
        res['spmm_A_F%d' % F] = {'ms': med, 'min_ms': mn, 'edges_per_s': E / med * 1e3,
                                 'alg_GBps': bytes_alg / med / 1e6, 'frac_8TBps': bytes_alg / med / 1e6 / 8000}
        print('spmm A F=%d: %.3f ms (min %.3f)  %.2e edges/s  alg %.0f GB/s (%.1f%% of 8 TB/s)' %
              (F, med, mn, E / med * 1e3, bytes_alg / med / 1e6, bytes_alg / med / 1e6 / 80), flush=True)
    if do('spmm'):
      dX = ops.CSR(X, dev, args.long, args.chunk)
      W0 = ops.DMat.from_numpy((rng.randn(s.V, 300) * 0.05).astype(np.float32), dev)
      out = ops.DMat(N, 300, dev)
      med, mn = timeit(lambda: ops.spmm(dX, W0, out=out), args.reps)
      res['spmm_X'] = {'ms': med}
      print('spmm X.W0: %.3f ms' % med, flush=True)
      import scipy.sparse as sps
      dXt = ops.CSR(sps.csr_matrix(X.T), dev, args.long, args.chunk)
      print('Xt long rows', dXt.n_long_rows, 'chunks', dXt.n_chunks, flush=True)
      G = ops.DMat.from_numpy(rng.randn(N, 300).astype(np.float32), dev)
      outw = ops.DMat(s.V, 300, dev)
      med, mn = timeit(lambda: ops.spmm(dXt, G, out=outw), args.reps)
      res['spmm_Xt'] = {'ms': med}
      print('spmm Xt.G: %.3f ms' % med, flush=True)
    # GEMMs
    H = ops.DMat.from_numpy(rng.randn(N, 300).astype(np.float32), dev)
    for Fo in ((300, 600, s.C) if do('gemm') else ()):
        W = ops.DMat.from_numpy((rng.randn(300, Fo) * 0.05).astype(np.float32), dev)
        Z = ops.DMat(N, Fo, dev)
        med, mn = timeit(lambda: ops.gemm(H, W, out=Z, precision=args.precision), args.reps)
        fl = 2.0 * N * 300 * Fo
        res['gemm_nn_%d' % Fo] = {'ms': med, 'TFLOPs': fl / med / 1e9}
        print('gemm NN N x300x%d: %.3f ms  %.1f TF' % (Fo, med, fl / med / 1e9), flush=True)
        dH = ops.DMat(N, 300, dev)
        med, mn = timeit(lambda: ops.gemm(Z, W, out=dH, transB=True, precision=args.precision), args.reps)
        res['gemm_nt_%d' % Fo] = {'ms': med, 'TFLOPs': fl / med / 1e9}
        print('gemm NT (dH) %d: %.3f ms  %.1f TF' % (Fo, med, fl / med / 1e9), flush=True)
        dW = ops.DMat(300, Fo, dev)
        med, mn = timeit(lambda: ops.gemm(H, Z, out=dW, transA=True), args.reps)
        res['gemm_tn_%d' % Fo] = {'ms': med, 'TFLOPs': fl / med / 1e9}
        print('gemm TN (dW) %d: %.3f ms  %.1f TF' % (Fo, med, fl / med / 1e9), flush=True)
    # streaming kernels
    if not do('stream'):
        return
    G = ops.DMat.from_numpy(rng.randn(N, 300).astype(np.float32), dev)
    T = ops.DMat.from_numpy(rng.rand(N, 300).astype(np.float32), dev)
    Hc = ops.DMat.from_numpy(rng.randn(N, 300).astype(np.float32), dev)
    o = ops.DMat(N, 300, dev)
    med, mn = timeit(lambda: ops.highway_fwd(T, Hc, H, out=o), args.reps)
    print('highway_fwd: %.3f ms  %.0f GB/s' % (med, 4 * N * 300 * 4 / med / 1e6), flush=True)
    res['highway_fwd'] = {'ms': med, 'GBps': 4 * N * 300 * 4 / med / 1e6}
    a, b, c = ops.DMat(N, 300, dev), ops.DMat(N, 300, dev), ops.DMat(N, 300, dev)
    med, mn = timeit(lambda: ops.highway_bwd(G, T, Hc, H, a, b, c), args.reps)
    print('highway_bwd: %.3f ms  %.0f GB/s' % (med, 7 * N * 300 * 4 / med / 1e6), flush=True)
    res['highway_bwd'] = {'ms': med, 'GBps': 7 * N * 300 * 4 / med / 1e6}
    dbS, dbU = torch.zeros(300, device=dev), torch.zeros(300, device=dev)
    med, mn = timeit(lambda: ops.highway_bwd(G, T, Hc, H, a, b, c, dbS, dbU), args.reps)
    print('highway_bwd + column sums: %.3f ms  %.0f GB/s' % (med, 7 * N * 300 * 4 / med / 1e6), flush=True)
    res['highway_bwd_colsum'] = {'ms': med, 'GBps': 7 * N * 300 * 4 / med / 1e6}
    med, mn = timeit(lambda: ops.act_bwd_colsum(G, Hc, ops.ACT_TANH, dbS, out=a), args.reps)
    print('act_bwd + column sums: %.3f ms  %.0f GB/s' % (med, 3 * N * 300 * 4 / med / 1e6), flush=True)
    res['act_bwd_colsum'] = {'ms': med, 'GBps': 3 * N * 300 * 4 / med / 1e6}
    med, mn = timeit(lambda: ops.colsum(G), args.reps)
    print('colsum: %.3f ms  %.0f GB/s' % (med, N * 300 * 4 / med / 1e6), flush=True)
    res['colsum'] = {'ms': med}
    L = ops.DMat.from_numpy(rng.randn(N, s.C).astype(np.float32), dev)
    P = ops.DMat(N, s.C, dev)
    am = torch.zeros(N, dtype=torch.int32, device=dev)
    med, mn = timeit(lambda: ops.softmax_rows(L, out=P, argmax=am), args.reps)
    print('softmax: %.3f ms  %.0f GB/s' % (med, 2 * N * s.C * 4 / med / 1e6), flush=True)
    res['softmax'] = {'ms': med}
    os.makedirs('gpurun_out', exist_ok=True)
    with open('gpurun_out/bench_kernels_%s.json' % args.shape, 'w') as f:
        json.dump(res, f, indent=1)


if __name__ == '__main__':
    main()
