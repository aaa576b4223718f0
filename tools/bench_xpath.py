"""X-path micro-benchmark at TwitterUS shape (run on the MI355X): X.W0 and X^T.dS0 in their formulations.
   python tools/bench_xpath.py [--reps 20]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from geographconv_amd import ops, synth  # noqa: E402
from tools.bench_kernels import timeit  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--shape', default='twus')
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    s = synth.SHAPES[args.shape]
    X = synth.bow_x(s.N, s.V, s.mean_nnz)
    x = ops.SparseOperand.from_scipy(X, dev)
    F = 300
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

    show('X.W0   one CSR gather kernel', timeit(lambda: ops.spmm(x.fwd, W, out=out, bias=b, act=1), args.reps)[0], alg_fwd)
    for slab in (0, 64, 128):
        ops.X_FWD_SLAB = slab
        show('X.W0   dense head GEMM + tail (slab %d)' % slab,
             timeit(lambda: ops.spmm_x(x, W, out=out, bias=b, act=1), args.reps)[0], alg_fwd)
    ops.X_FWD_SLAB = 0
    Wh = ops.DMat(x.head_dense.F, F, dev)
    show('   head GEMM alone  N x %d x %d' % (x.head_dense.F, F), timeit(lambda: ops.gemm(x.head_dense, Wh, out=out), args.reps)[0], alg_fwd)
    show('   tail accumulate alone', timeit(lambda: ops.spmm(x.fwd_tail, W, out=out, bias=b, act=1, accumulate=True), args.reps)[0], alg_fwd)
    for c in (64, 128):
        show('   tail, one %d-column slab' % c, timeit(lambda: ops._spmm_cols(x.fwd_tail, W, out, b, 1, 0, c), args.reps)[0], alg_fwd)

    for name, g in (('ld 320', G), ('ld 300', G300)):
        ops.XT_MIN_NNZ = 0
        show('X^T.dS0  head GEMM + document-blocked tail (%s)' % name, timeit(lambda: ops.spmm_t(x, g, out=dW), args.reps)[0], alg_bwd)
        ops.XT_MIN_NNZ = 1 << 60
        show('X^T.dS0  head GEMM + row-gather tail (%s)' % name, timeit(lambda: ops.spmm_t(x, g, out=dW), args.reps)[0], alg_bwd)
    ops.XT_MIN_NNZ = 0
    plan = x.xt_plan(F)
    w = ops._ws_for(dev).get(plan.ws_bytes)
    from geographconv_amd import _ffi
    show('   document-blocked tail alone', timeit(lambda: _ffi.check(_ffi.lib().geogcn_xt_dot_f32(
        plan._h, ops._p(x.bwd.colidx), ops._p(x.bwd.val), ops._p(G.t), G.ld, ops._p(dW.t), dW.ld, ops._p(w), w.numel(),
        ops._stream())), args.reps)[0], alg_bwd)
    show('   head GEMM alone (transA)', timeit(lambda: ops.gemm(x.head_dense, G, transA=True), args.reps)[0], alg_bwd)


if __name__ == '__main__':
    main()
