"""Generates the committed golden vectors under tests/golden/.

The reference holds NO golden vectors for this path and cannot be imported here (Theano/Lasagne
absent -- SURVEY.md §8c), so these fixtures are produced by the repo's own CPU restatement
(oracle/gcn_oracle.py, "parity unpinned").  They pin the oracle against silent drift and give the
GPU tests small fixed cases (inputs AND expected outputs travel; nothing is regenerated on the
GPU box).   python tests/golden/make_golden.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from geographconv_amd import synth  # noqa: E402
from oracle import gcn_oracle as O  # noqa: E402


def case(name, N, V, C, hid, highway, p, reg, seed):
    A, X, Y = synth.small_graph(N, 4.0, V, 8, C, seed=seed, empty_rows=2)
    params = O.random_params(V, hid, C, highway, seed=seed + 1, scale=0.5)
    tr = np.arange(0, int(N * 0.6), dtype=np.int32)
    dev = np.arange(int(N * 0.6), int(N * 0.8), dtype=np.int32)
    te = np.arange(int(N * 0.8), N, dtype=np.int32)
    mask = (np.random.RandomState(seed + 2).rand(N, hid[0]) < (1 - p)).astype(np.uint8) if p > 0 else \
        np.ones((N, hid[0]), np.uint8)
    st = O.AdamState(params)
    cur = [q.copy() for q in params]
    out = {}
    for step in range(2):
        new, outs, grads = O.f_train(cur, st, X, Y[tr], Y[dev], A, tr, dev, hid, highway, p,
                                     mask.astype(np.float32), reg)
        out['step%d_scalars' % step] = np.array(outs[:4], dtype=np.float64)
        out['step%d_P' % step] = outs[4]
        for i, g in enumerate(grads):
            out['step%d_grad%d' % (step, i)] = g
        for i, q in enumerate(new):
            out['step%d_param%d' % (step, i)] = q
        cur = new
    pred, probs = O.f_val(cur, X, A, te, hid, highway)
    out['val_pred'] = pred
    out['val_probs'] = probs
    np.savez_compressed(
        os.path.join(HERE, name + '.npz'),
        A_indptr=A.indptr, A_indices=A.indices, A_data=A.data, X_indptr=X.indptr, X_indices=X.indices,
        X_data=X.data, N=N, V=V, C=C, hid=np.array(hid), highway=highway, p=p, reg=reg, Y=Y, tr=tr, dev=dev, te=te,
        mask=mask, n_params=len(params), **{'param%d' % i: q for i, q in enumerate(params)}, **out)
    print(name, 'written; loss', out['step0_scalars'][0], 'nnzA', A.nnz)


if __name__ == '__main__':
    case('tiny_highway', 96, 40, 5, [12, 12, 12], True, 0.5, 0.0, 0)
    case('tiny_plain_reg', 80, 30, 4, [8, 12, 6], False, 0.0, 1e-3, 10)
    case('tiny_odd_widths', 70, 25, 7, [10, 10], True, 0.2, 0.0, 20)
