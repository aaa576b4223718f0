"""SURVEY.md 8(d) timing protocol for the dominant kernel: the F = 300 graph product back to back ("warm": whatever of Z,
the CSR and the output the 8 x 4 MB L2 / 256 MB Infinity Cache still hold from the previous launch) and after a 2 GB
streaming write that evicts both ("cold"); >= 10 warm-ups, >= 50 timed launches each, median / p10 / p90.
   python tools/spmm_cold_warm.py [--reps 50]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from geographconv_amd import ops, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=50)
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    s = synth.SHAPES['twus']
    A = synth.powerlaw_ahat(s.N, s.E_target)
    dA = ops.CSR(A, dev)
    Z = ops.DMat.empty(s.N, 300, dev, ld=ops.gather_ld(300))
    Z.t.normal_()
    out = ops.DMat(s.N, 300, dev)
    flush = torch.empty(512 * 1024 * 1024, dtype=torch.float32, device=dev)       # 2 GB
    alg = 8 * A.nnz + 4 * (s.N + 1) + 8 * s.N * 300

    def run(cold):
        ts = []
        for i in range(10 + args.reps):
            if cold:
                flush.fill_(float(i))
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            ops.spmm(dA, Z, out=out)
            b.record()
            b.synchronize()
            if i >= 10:
                ts.append(a.elapsed_time(b))
        ts.sort()
        return ts[len(ts) // 2], ts[len(ts) // 10], ts[(9 * len(ts)) // 10]

    for name, cold in (('warm (back to back)', False), ('cold (after a 2 GB streaming write)', True)):
        med, p10, p90 = run(cold)
        print('%-38s median %.3f ms  p10 %.3f  p90 %.3f   -> %.0f GB/s algorithmic = %.1f %% of 8 TB/s'
              % (name, med, p10, p90, alg / med / 1e6, alg / med / 1e6 / 80), flush=True)


if __name__ == '__main__':
    main()
