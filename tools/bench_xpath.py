"""X-path micro-benchmark at TwitterUS shape (run on the MI355X): X.W0 and X^T.dS0 in their formulations.
   python tools/bench_xpath.py [--reps 20]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from geographconv_amd import ops, synth, tuning  # noqa: E402
from tools.bench_kernels import timeit  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--shape', default='twus')
    ap.add_argument('--only', default='all', choices=['all', 'fwd', 'xt'])
    ap.add_argument('--F', type=int, default=300)
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    s = synth.SHAPES[args.shape]
    X = synth.bow_x(s.N, s.V, s.mean_nnz)
    x = ops.SparseOperand.from_scipy(X, dev)
    F = args.F
    rng = np.random.RandomState(1)
    W = ops.DMat.from_numpy((rng.randn(s.V, F) * 0.05).astype(np.float32), dev)
    b = torch.zeros(ops.pad4(F), device=dev)
    G = ops.DMat.empty(s.N, F, dev, ld=ops.gather_ld(F))
    G.t.normal_()
    G300 = ops.DMat.from_numpy(G.numpy(), dev)
    out = ops.DMat(s.N, F, dev)
    dW = ops.DMat(s.V, F, dev)
    nnzX, nt = X.nnz, x.bwd.nnz
    print('X: nnz %d, head %d columns, tail nnz %d (%.0f %%)' % (nnzX, x.head_dense.F, nt, 100.0 * nt / nnzX))
    alg_fwd = 8 * nnzX + 4 * (s.N + 1) + 4 * s.V * F + 4 * s.N * F
    alg_bwd = 8 * nnzX + 4 * s.N * F + 4 * s.V * F

    def show(name, ms, alg):
        print('%-52s %.3f ms   alg %.0f GB/s (%.1f %% of 8 TB/s)' % (name, ms, alg / ms / 1e6, alg / ms / 1e6 / 80), flush=True)

    if args.only in ('all', 'fwd'):
      cap = int(__import__('geographconv_amd._ffi', fromlist=['lib']).lib().geogcn_spmm_hot_capacity(F))
      hot = ops.HotCSR(x.fwd, X.data, cap)
      print('hot rows in LDS: %d -> %.1f %% of the stored entries' % (hot.n_hot, 100 * hot.hot_fraction))
      show('X.W0   hot rows of W0 in LDS (%d rows)' % hot.n_hot, timeit(lambda: ops.spmm_hot(hot, W, out=out, bias=b, act=1), args.reps)[0], alg_fwd)
      for nh in (32, 64):
          h2 = ops.HotCSR(x.fwd, X.data, nh)
          show('X.W0   hot rows of W0 in LDS (%d rows, %.0f %%)' % (nh, 100 * h2.hot_fraction),
               timeit(lambda: ops.spmm_hot(h2, W, out=out, bias=b, act=1), args.reps)[0], alg_fwd)
      show('X.W0   one CSR gather kernel', timeit(lambda: ops.spmm(x.fwd, W, out=out, bias=b, act=1), args.reps)[0], alg_fwd)
    if args.only == 'fwd':
        return
    for name, g in (('ld 320', G), ('ld 300', G300)):
        tuning.XT_MIN_NNZ = 0
        show('X^T.dS0  head GEMM + document-blocked tail (%s)' % name, timeit(lambda: ops.spmm_t(x, g, out=dW), args.reps)[0], alg_bwd)
        tuning.XT_MIN_NNZ = 1 << 60
        show('X^T.dS0  head GEMM + row-gather tail (%s)' % name, timeit(lambda: ops.spmm_t(x, g, out=dW), args.reps)[0], alg_bwd)
    tuning.XT_MIN_NNZ = 0
    plan = x.xt_plan(F)
    w = ops._ws_for(dev).get(plan.ws_bytes)
    from geographconv_amd import _ffi
    show('   document-blocked tail alone', timeit(lambda: _ffi.check(_ffi.lib().geogcn_xt_dot_f32(
        plan._h, ops._p(x.bwd.colidx), ops._p(x.bwd.val), ops._p(G.t), G.ld, ops._p(dW.t), dW.ld, ops._p(w), w.numel(),
        ops._stream())), args.reps)[0], alg_bwd)
    show('   head GEMM alone (transA)', timeit(lambda: ops.gemm(x.head_dense, G, transA=True), args.reps)[0], alg_bwd)
    # head size: the cost model's choice (above) against forced sizes (180 = what the 3.5 % density rule of rounds 1-2 took)
    del plan
    for k in (160, 256, 320, 384, 480):
        tuning.DENSE_HEAD_SIZES = (k,)
        xk = ops.SparseOperand.from_scipy(X, dev)
        tuning.XT_MIN_NNZ = 0
        show('X^T.dS0  head of %d columns + tail (%d entries)' % (xk.head_dense.F if xk.head_dense is not None else 0, xk.bwd.nnz),
             timeit(lambda: ops.spmm_t(xk, G, out=dW), args.reps)[0], alg_bwd)
        del xk


if __name__ == '__main__':
    main()
