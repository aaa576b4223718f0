"""Reference-generated golden vectors: the way from "parity unpinned" to "pinned".

    python tests/golden/make_reference_golden.py [--reference /root/reference] [--cases tiny_highway ...]

Runs ONLY where Theano (1.0.x) and Lasagne (0.2.dev1 / git master, as the reference's README asks) are importable --
not in the build container of this repository and never on the GPU box.  It imports the REFERENCE's own
``gcnmodel.py`` from ``--reference``, builds its ``GraphConv`` with the reference's own ``build_model`` (so the
``theano.function``s compiled at /root/reference/gcnmodel.py:409-411 are the ones that run), and feeds it the inputs of the
committed fixtures ``tests/golden/<case>.npz`` (A_hat, X, Y, index vectors, explicit weights, dropout keep-mask).  Outputs
go to ``tests/golden/ref_<case>.npz`` in the SAME schema as the oracle-made fixtures (step{0,1}_scalars / _P / _grad<i> /
_param<i>, val_pred, val_probs), so ``tests/test_golden.py`` compares the oracle -- and on the GPU the HIP path --
against Theano's numbers whenever those files are present, and reports "parity unpinned" when they are not.

Only data is written: inputs and expected outputs.  Nothing of the reference's source is copied.

Two things the reference does not expose and how they are obtained without editing it:
  * explicit weights: ``lasagne.layers.set_all_param_values(clf.l_out, params)`` after ``build_model`` (the compiled
    functions read the shared variables, so the new values are what f_train uses; Adam's m / v / t start at zero as in a
    fresh model);
  * a KNOWN dropout mask: Theano's MRG31k3p stream cannot be reproduced elsewhere, so while ``build_model`` runs,
    ``lasagne.layers.dropout`` (the name gcnmodel.py:357 calls) is replaced by a layer that multiplies by the fixture's
    constant keep-mask and 1/(1-p) when ``deterministic=False`` -- exactly ``DropoutLayer.get_output_for`` with the random
    draw replaced by a constant -- and is the identity when ``deterministic=True``;
  * gradients: f_train returns none, so a separate ``theano.function`` evaluates ``theano.grad(clf.train_loss, params)``
    on the same graph before each step.
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np
import scipy.sparse as sps

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = ['tiny_highway', 'tiny_plain_reg', 'tiny_odd_widths']


def load_fixture(name):
    z = np.load(os.path.join(HERE, name + '.npz'))
    N, V = int(z['N']), int(z['V'])
    A = sps.csr_matrix((z['A_data'], z['A_indices'], z['A_indptr']), shape=(N, N)).astype('float32')
    X = sps.csr_matrix((z['X_data'], z['X_indices'], z['X_indptr']), shape=(N, V)).astype('float32')
    params = [np.asarray(z['param%d' % i], dtype='float32') for i in range(int(z['n_params']))]
    cfg = dict(N=N, V=V, C=int(z['C']), hid=[int(h) for h in z['hid']], highway=bool(z['highway']), p=float(z['p']),
               reg=float(z['reg']))
    return z, A, X, params, cfg


def run_case(name, R, theano, lasagne):
    import theano.tensor as T
    z, A, X, params, cfg = load_fixture(name)
    mask = theano.shared(np.asarray(z['mask'], dtype='float32'), name='injected_keep_mask')

    class InjectedDropout(lasagne.layers.Layer):
        """lasagne.layers.DropoutLayer (rescale=True) with the Bernoulli draw replaced by a constant keep-mask."""

        def __init__(self, incoming, p=0.5, **kwargs):
            super(InjectedDropout, self).__init__(incoming, **kwargs)
            self.p = p

        def get_output_for(self, input, deterministic=False, **kwargs):
            if deterministic or self.p == 0:
                return input
            return input * mask / T.constant(1.0 - self.p, dtype=input.dtype)

    real_dropout = lasagne.layers.dropout
    lasagne.layers.dropout = lambda incoming, p=0.5, **kw: InjectedDropout(incoming, p=p, **kw)
    try:
        clf = R.GraphConv(input_size=cfg['V'], output_size=cfg['C'], hid_size_list=cfg['hid'], regul_coef=cfg['reg'],
                          drop_out=cfg['p'], highway=cfg['highway'])
        clf.build_model(A, use_text=True, use_labels=True, seed=77)
    finally:
        lasagne.layers.dropout = real_dropout
    ref_params = lasagne.layers.get_all_params(clf.l_out, trainable=True)
    shapes = [tuple(p.get_value().shape) for p in ref_params]
    assert shapes == [tuple(q.shape) for q in params], \
        "parameter order / shapes differ from the fixture: %r vs %r" % (shapes, [q.shape for q in params])
    lasagne.layers.set_all_param_values(clf.l_out, params)

    f_grad = theano.function([clf.X_sym, clf.train_y_sym, clf.A_sym, clf.train_indices_sym],
                             theano.grad(clf.train_loss, ref_params), on_unused_input='warn')
    Y = np.asarray(z['Y'])
    tr, dev, te = (np.asarray(z[k]).astype('int64') for k in ('tr', 'dev', 'te'))
    y_tr, y_dev = Y[tr].astype('int64'), Y[dev].astype('int64')
    out = {}
    for step in range(2):
        grads = f_grad(X, y_tr, A, tr)
        l_tr, a_tr, l_dev, a_dev, P = clf.f_train(X, y_tr, y_dev, A, tr, dev)
        out['step%d_scalars' % step] = np.array([l_tr, a_tr, l_dev, a_dev], dtype=np.float64)
        out['step%d_P' % step] = np.asarray(P, dtype=np.float32)
        for i, g in enumerate(grads):
            out['step%d_grad%d' % (step, i)] = np.asarray(g, dtype=np.float32)
        for i, q in enumerate(lasagne.layers.get_all_param_values(clf.l_out)):
            out['step%d_param%d' % (step, i)] = np.asarray(q, dtype=np.float32)
    pred, probs = clf.predict(X, A, te)
    out['val_pred'] = np.asarray(pred)
    out['val_probs'] = np.asarray(probs, dtype=np.float32)
    keep = {k: z[k] for k in z.files if not (k.startswith('step') or k.startswith('val_'))}      # the inputs, unchanged
    path = os.path.join(HERE, 'ref_' + name + '.npz')
    np.savez_compressed(path, made_by='make_reference_golden.py', theano_version=str(theano.__version__),
                        lasagne_version=str(lasagne.__version__), floatX=str(theano.config.floatX), **keep, **out)
    print('%s written: train loss %.6f (oracle-made fixture: %.6f)' % (path, out['step0_scalars'][0], float(z['step0_scalars'][0])))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reference', default='/root/reference', help='checkout of afshinrahimi/geographconv')
    ap.add_argument('--cases', nargs='+', default=CASES)
    args = ap.parse_args()
    os.environ.setdefault('THEANO_FLAGS', 'floatX=float32,device=cpu')           # the reference's CPU path (README)
    try:
        import theano
        import lasagne
    except ImportError as e:
        raise SystemExit("this generator needs Theano + Lasagne (requirements.txt of the reference): %s -- parity stays "
                         "UNPINNED until it has been run somewhere that has them" % e)
    sys.path.insert(0, args.reference)
    import gcnmodel as R                      # the reference's module, imported, never copied
    assert os.path.abspath(R.__file__).startswith(os.path.abspath(args.reference)), R.__file__
    for name in args.cases:
        run_case(name, R, theano, lasagne)


if __name__ == '__main__':
    main()
