"""The graph product A_hat . Z (F = 300) on graphs with and without locality, under each node reordering -- run plain
for timings, or under `rocprofv3 --pmc ...` for the L2 / fabric counters (tools/pmc_locality_summary.py reads the passes).
   python tools/spmm_locality.py [--cases pinned,pinned+degree,pinned+lpa,sbm,sbm+rcm,sbm+lpa] [--reps 10]
Each case launches the SpMM `reps` times; the case order is printed so that the counters of launch k can be attributed."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from geographconv_amd import graph, ops, synth  # noqa: E402
from tools.bench_kernels import timeit  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--cases', default='pinned,pinned+degree,pinned+lpa,sbm,sbm+degree,sbm+rcm,sbm+bfs,sbm+lpa')
    ap.add_argument('--reps', type=int, default=10)
    ap.add_argument('--F', type=int, default=300)
    ap.add_argument('--communities', type=int, default=synth.TWUS_SBM_COMMUNITIES)
    ap.add_argument('--out', default='gpurun_out/spmm_locality.json')
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    s = synth.SHAPES['twus']
    base = {}
    res = []
    for case in args.cases.split(','):
        name, _, method = case.partition('+')
        if name not in base and name == 'band':
            # every neighbour within +-600 rows: the gather's ideal case (an upper bound for what reordering can give)
            import scipy.sparse as sps
            rng = np.random.RandomState(0)
            k = 24
            rows = np.repeat(np.arange(s.N), k)
            cols = np.clip(rows + rng.randint(-600, 601, size=len(rows)), 0, s.N - 1)
            B0 = sps.coo_matrix((np.ones(len(rows), np.float32), (rows, cols)), shape=(s.N, s.N)).tocsr()
            B0.data[:] = 1.0 / k
            B0.sort_indices()
            base[name] = B0
            print('[band generated: nnz %d]' % B0.nnz, flush=True)
        if name not in base:
            t0 = time.time()
            base[name] = synth.powerlaw_ahat(s.N, s.E_target) if name == 'pinned' else synth.community_ahat(s.N, s.E_target, args.communities)
            print('[%s generated in %.1f s: nnz %d]' % (name, time.time() - t0, base[name].nnz), flush=True)
        A = base[name]
        t0 = time.time()
        ro = graph.reordering(A, method or None)
        B = A if ro is None else ro.matrix(A)
        t_re = time.time() - t0
        loc = graph.locality_profile(B)
        dA = ops.CSR(B, dev)
        Z = ops.DMat.empty(s.N, args.F, dev, ld=ops.gather_ld(args.F))
        Z.t.normal_()
        out = ops.DMat(s.N, args.F, dev)
        med, mn = timeit(lambda: ops.spmm(dA, Z, out=out), args.reps)
        alg = 8 * B.nnz + 4 * (s.N + 1) + 8 * s.N * args.F
        r = {'case': case, 'nnz': int(B.nnz), 'ms': med, 'min_ms': mn, 'alg_GBps': alg / med / 1e6, 'frac_of_8TBps': alg / med / 1e6 / 8000,
             'edges_within_L2_window': loc['within_window'], 'reorder_seconds': t_re, 'launches': args.reps + 3}
        res.append(r)
        print('%-14s %.3f ms  alg %.0f GB/s = %.1f %% of 8 TB/s   edges within an L2 window of their row: %.1f %%   (reorder %.1f s)'
              % (case, med, r['alg_GBps'], 100 * r['frac_of_8TBps'], 100 * loc['within_window'], t_re), flush=True)
        del dA, Z, out
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(res, open(args.out, 'w'), indent=1)


if __name__ == '__main__':
    main()
