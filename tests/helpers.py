import os

import numpy as np
import scipy.sparse as sps

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
CASES = ['tiny_highway', 'tiny_plain_reg', 'tiny_odd_widths']


def load_case(name):
    z = np.load(os.path.join(GOLDEN, name + '.npz'))
    N, V = int(z['N']), int(z['V'])
    A = sps.csr_matrix((z['A_data'], z['A_indices'], z['A_indptr']), shape=(N, N))
    X = sps.csr_matrix((z['X_data'], z['X_indices'], z['X_indptr']), shape=(N, V))
    params = [z['param%d' % i] for i in range(int(z['n_params']))]
    cfg = dict(N=N, V=V, C=int(z['C']), hid=[int(h) for h in z['hid']], highway=bool(z['highway']), p=float(z['p']),
               reg=float(z['reg']))
    return z, A, X, params, cfg


def make_clf(cfg, params, device=None, comm=None):
    """GraphConv with explicit weights (fixtures never rely on the initialisers)."""
    from geographconv_amd.gcnmodel import GraphConv
    from geographconv_amd.nn import layers as L
    clf = GraphConv(cfg['V'], cfg['C'], cfg['hid'], cfg['reg'], cfg['p'], highway=cfg['highway'], device=device,
                    comm=comm)
    clf.build_model(None, seed=77)
    L.set_all_param_values(clf.l_out, params)
    return clf
