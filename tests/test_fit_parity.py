"""GraphConv.fit (reference gcnmodel.py:418-450) against the oracle's line-for-line restatement of that
loop (oracle.fit): which epoch is the best one, at which epoch training stops, WHICH parameter values are snapshotted (those
after the best epoch's update, gcnmodel.py:437 runs after f_train's updates) and that they are what fit() leaves in the model.

CPU part: the stopping rule alone (`_DevLossWatch`) on scripted dev-loss sequences.  GPU part: whole fits through the C ABI.
Parity unpinned, as everywhere: the oracle is this repository's restatement of the reference."""
import gzip
import math
import os
import pickle

import numpy as np
import pytest

from oracle import gcn_oracle as O
from tests.helpers import load_case


# --------------------------------------------------------------------------------------------
# the rule on scripted sequences (no GPU)
# --------------------------------------------------------------------------------------------
def _watch_run(losses, max_down, n_epochs=None):
    """Drive _DevLossWatch exactly as GraphConv.fit does; -> (best_epoch, stop_epoch, stopped_early, n_down per epoch)."""
    from geographconv_amd.gcnmodel import _DevLossWatch
    w = _DevLossWatch(max_down)
    best, downs, stopped, epoch = -1, [], False, -1
    for epoch, l in enumerate(losses if n_epochs is None else losses[:n_epochs]):
        if w.update(l, 0.5):
            best = epoch
        downs.append(w.n_down)
        if w.exhausted(epoch):
            stopped = True
            break
    return best, epoch, stopped, downs


def _oracle_run(losses, max_down, n_epochs=None):
    it = iter(losses)

    def step(params, st):
        return params, (0.0, 0.0, next(it), 0.5)
    r = O.fit([np.zeros(1)], None, step, n_epochs=len(losses) if n_epochs is None else n_epochs, max_down=max_down)
    return r['best_epoch'], r['stop_epoch'], r['stopped_early'], [h[4] for h in r['history']]


SCRIPTED = {
    'strict_improvement': [5.0, 4.0, 3.0, 2.0, 1.0, 0.5, 0.4, 0.3, 0.2, 0.1, 0.05, 0.01],
    'ties_do_not_improve': [2.0, 2.0, 2.0, 2.0, 2.0, 2.0, 2.0, 2.0, 2.0, 2.0, 2.0, 2.0],
    'worse_from_the_start': [1.0, 1.1, 1.2, 1.3, 1.4, 1.5, 1.6, 1.7, 1.8, 1.9, 2.0, 2.1, 2.2, 2.3, 2.4, 2.5, 2.6, 2.7],
    'down_then_up': [3.0, 2.5, 2.2, 2.1, 2.05, 2.06, 2.07, 2.2, 2.3, 2.4, 2.5, 2.6, 2.7, 2.8, 2.9, 3.0, 3.1, 3.2],
    'zigzag_resets_the_count': [3.0, 3.1, 3.2, 2.9, 3.0, 3.1, 2.8, 2.9, 3.0, 2.7, 2.8, 2.9, 3.0, 3.1, 3.2, 3.3, 3.4, 3.5, 3.6],
    'nan_never_improves': [2.0, float('nan'), 1.5, float('nan'), float('nan'), float('nan'), float('nan'), float('nan'),
                           float('nan'), float('nan'), float('nan'), float('nan'), float('nan')],
    'nan_first': [float('nan')] * 12,
    'inf_and_huge': [float('inf'), 1e19, 9.0e18, 1.0, 2.0, 3.0, 4.0, 5.0, 6.0, 7.0, 8.0, 9.0, 10.0],
    'late_recovery_inside_the_grace_period': [1.0, 2.0, 2.0, 2.0, 2.0, 0.5, 2.0, 2.0, 2.0, 2.0, 2.0, 2.0, 2.0, 2.0],
}


@pytest.mark.parametrize('name', sorted(SCRIPTED))
@pytest.mark.parametrize('max_down', [0, 1, 2, 3, 10])
def test_dev_loss_watch_follows_the_reference_rule_on_scripted_sequences(name, max_down):
    seq = SCRIPTED[name]
    assert _watch_run(seq, max_down) == _oracle_run(seq, max_down)


def test_dev_loss_watch_known_answers():
    """Hand-checked against gcnmodel.py:434-447 (stop iff n_down > max_down AND n > 2 * max_down)."""
    up = SCRIPTED['worse_from_the_start']
    # best stays epoch 0; n_down = n.  max_down 3: n_down > 3 from n = 4, n > 6 from n = 7  => stops at epoch 7
    assert _watch_run(up, 3)[:3] == (0, 7, True)
    # max_down 2: n_down > 2 at n = 3 but n must exceed 4 => epoch 5
    assert _watch_run(up, 2)[:3] == (0, 5, True)
    # max_down 0: n_down > 0 at n = 1, n > 0 => epoch 1
    assert _watch_run(up, 0)[:3] == (0, 1, True)
    # `n_down > max_down while epoch <= 2 * max_down` does NOT stop: at epoch 6 n_down is 6 > 3 and 6 > 6 is false
    assert _watch_run(up, 3, n_epochs=7)[:3] == (0, 6, False)
    # ties are not improvements (strict <), so a constant loss stops like a rising one
    assert _watch_run(SCRIPTED['ties_do_not_improve'], 3)[:3] == (0, 7, True)
    # the count is reset by every improvement: minimum at epoch 9, then 4 worse epochs => stop at 13
    assert _watch_run(SCRIPTED['zigzag_resets_the_count'], 3)[:3] == (9, 13, True)
    # NaN < x is False: never best, always "down"
    assert _watch_run(SCRIPTED['nan_never_improves'], 3)[:3] == (2, 7, True)
    assert _watch_run(SCRIPTED['nan_first'], 3)[0] == -1
    # sys.maxsize is the initial best (gcnmodel.py:422): inf and 1e19 do not beat it, 9.0e18 does
    assert _watch_run(SCRIPTED['inf_and_huge'], 10)[0] == 3
    assert _watch_run(SCRIPTED['inf_and_huge'][:3], 10)[0] == 2
    # an improvement during the 2 * max_down grace period counts
    assert _watch_run(SCRIPTED['late_recovery_inside_the_grace_period'], 3)[:3] == (5, 9, True)


def test_dev_loss_watch_random_sequences():
    rng = np.random.RandomState(0)
    for trial in range(300):
        n = rng.randint(1, 40)
        seq = list(np.round(rng.rand(n) * 3, rng.randint(0, 3)))            # coarse rounding => many exact ties
        if trial % 7 == 0:
            seq[rng.randint(n)] = float('nan')
        md = int(rng.randint(0, 6))
        assert _watch_run(seq, md) == _oracle_run(seq, md), (seq, md)


def test_oracle_fit_snapshots_the_parameters_after_the_update():
    """oracle.fit itself: the snapshot is what get_all_param_values returns AFTER f_train applied its updates."""
    losses = iter([3.0, 2.0, 2.5, 2.6, 2.7, 2.8])

    def step(params, st):
        return [params[0] + 1.0], (0.0, 0.0, next(losses), 0.0)
    r = O.fit([np.zeros(2)], None, step, n_epochs=6, max_down=1)
    assert r['best_epoch'] == 1 and r['stop_epoch'] == 3 and r['stopped_early']
    assert np.array_equal(r['best_params'][0], [2.0, 2.0])                  # two updates have run by the end of epoch 1
    assert np.array_equal(r['last_params'][0], [4.0, 4.0])


# --------------------------------------------------------------------------------------------
# whole fits through the HIP path
# --------------------------------------------------------------------------------------------
LOSS_RTOL = 2e-5          # per-epoch dev / train loss, HIP vs oracle, over up to ~130 Adam steps of drift
STEP = 2e-3               # Adam's learning rate (gcnmodel.py:407): an update moves an entry by <= ~lr


def _oracle_fit(params, X, A, Y, tr, dev, hid, highway, p, mask, reg, n_epochs, max_down, keep_all=False):
    st = O.AdamState(params)
    trail = []

    def step(cur, st):
        new, outs, _ = O.f_train(cur, st, X, Y[tr], Y[dev], A, tr, dev, hid, highway, p, mask, reg)
        if keep_all:
            trail.append(new)
        return new, outs
    r = O.fit([q.copy() for q in params], st, step, n_epochs=n_epochs, max_down=max_down)
    r['trail'] = trail
    return r


def _check_against_oracle(clf, r, params0, n_epochs, max_down):
    """Same decisions, same curve, same snapshot.  A decision of the oracle whose margin is inside the loss tolerance is
    arbitrated the way the argmax tests arbitrate ties: the HIP path may then differ, but only within that band."""
    from geographconv_amd.nn import layers as L
    hist = clf.fit_history
    ref = r['history']
    # (1) the rule applied to the HIP path's OWN dev losses gives the HIP path's decisions
    #     (padded with +inf up to n_epochs: stopping an epoch too early or too late both show)
    own = _oracle_run([h[2] for h in hist] + [math.inf] * (n_epochs - len(hist)), max_down)
    assert (clf.best_epoch, len(hist) - 1, len(hist) < n_epochs) == own[:3]
    assert [h[4] for h in hist] == own[3]
    # (2) the curve: every epoch both ran
    for e, (h, q) in enumerate(zip(hist, ref)):
        assert abs(h[0] - q[0]) <= LOSS_RTOL * abs(q[0]), (e, h[0], q[0])
        assert abs(h[2] - q[2]) <= LOSS_RTOL * abs(q[2]), (e, h[2], q[2])
    # (3) same stop epoch / best epoch -- unless an oracle comparison was inside the tolerance band
    best, margins = math.inf, []
    for q in ref:
        margins.append(abs(q[2] - best))
        best = min(best, q[2])
    band = LOSS_RTOL * abs(ref[0][2])
    ambiguous = [e for e, m in enumerate(margins) if m <= 2 * band]
    if not ambiguous:
        assert len(hist) - 1 == r['stop_epoch'] and clf.best_epoch == r['best_epoch']
        assert [h[4] for h in hist] == [q[4] for q in ref]
    else:
        assert abs(clf.best_epoch - r['best_epoch']) <= len(ambiguous), (clf.best_epoch, r['best_epoch'], ambiguous)
        assert abs(ref[clf.best_epoch][2] - ref[r['best_epoch']][2]) <= 2 * band
    assert (len(hist) < n_epochs) == r['stopped_early'] or ambiguous
    # (4) the snapshot: parameters after the best epoch's update, restored into the model and exposed as best_params
    got = L.get_all_param_values(clf.l_out)
    assert all(np.array_equal(a, b) for a, b in zip(got, clf.best_params))
    want = r['trail'][clf.best_epoch]
    drift = STEP * 0.02 * (clf.best_epoch + 1) + 1e-7          # test_e2e's one-step bound, per epoch
    for i, (a, b) in enumerate(zip(got, want)):
        assert np.abs(a - b).max() <= drift, (i, np.abs(a - b).max(), drift)
    # ... and NOT the values before that update / of a neighbouring epoch: the snapshot is nearer to the oracle's epoch
    # `best` than to the epochs either side of it
    def dist(ps):
        return max(np.abs(a - b).max() for a, b in zip(got, ps))
    before = r['trail'][clf.best_epoch - 1] if clf.best_epoch > 0 else params0
    assert dist(want) < 0.25 * dist(before), (dist(want), dist(before))
    if clf.best_epoch + 1 < len(r['trail']):
        assert dist(want) < 0.25 * dist(r['trail'][clf.best_epoch + 1])
    return got


@pytest.mark.gpu
@pytest.mark.usefixtures('both_gemm_precisions')
@pytest.mark.parametrize('name,n_epochs,max_down,use_mask', [
    ('tiny_plain_reg', 40, 2, False),        # dev loss rises from the first epoch: best 0, stops by the 2 * max_down clause
    ('tiny_plain_reg', 40, 3, False),
    ('tiny_odd_widths', 40, 3, False),
    ('tiny_highway', 60, 3, False),          # falls for all 60 epochs: never stops, best = last
    ('tiny_highway', 400, 3, False),         # minimum near epoch 125, then early stopping
    ('tiny_highway', 400, 2, True),          # the same with the fixture's dropout mask injected (dev metrics: dropout-ON pass)
])
def test_fit_matches_the_oracle_loop_on_the_fixtures(name, n_epochs, max_down, use_mask):
    from tests.helpers import make_clf
    z, A, X, params, cfg = load_case(name)
    Y, tr, dev, te = z['Y'], z['tr'], z['dev'], z['te']
    p = cfg['p'] if use_mask else 0.0
    mask = z['mask'].astype(np.float32) if use_mask else None
    r = _oracle_fit(params, X, A, Y, tr, dev, cfg['hid'], cfg['highway'], p, mask, cfg['reg'], n_epochs, max_down, True)
    clf = make_clf(dict(cfg, p=p), params)
    if use_mask:
        clf.inject_dropout_mask(z['mask'])
    clf.fit(X, A, Y, tr, dev, n_epochs=n_epochs, max_down=max_down, verbose=False)
    assert clf.fitted
    got = _check_against_oracle(clf, r, params, n_epochs, max_down)
    # predict() after fit() runs on the restored snapshot: labels of the oracle's best parameters, outside the tie band
    pred, probs = clf.predict(X, A, te)
    rp, rows = O.f_val(r['trail'][clf.best_epoch], X, A, te, cfg['hid'], cfg['highway'])
    tol = 1e-3
    assert np.abs(probs - rows).max() <= tol
    top2 = np.sort(rows, axis=1)[:, -2:]
    clear = (top2[:, 1] - top2[:, 0]) > 2 * tol
    assert np.array_equal(pred[clear], rp[clear])
    # save -> load round trip in the reference's gzip-pickle list-of-arrays format (gcnmodel.py:459-470, data.py:28-34)
    import os
    import tempfile
    from geographconv_amd.gcnmain import dump_obj, load_obj
    with tempfile.TemporaryDirectory() as d:
        fn = os.path.join(d, 'model.pkl')
        clf.save(dump_obj, fn)
        with gzip.open(fn, 'rb') as f:
            raw = pickle.load(f)
        assert isinstance(raw, list) and all(np.array_equal(a, b) for a, b in zip(raw, got))
        clf2 = make_clf(dict(cfg, p=p), params)
        clf2.load(load_obj, fn)
        assert clf2.fitted
        pred2, probs2 = clf2.predict(X, A, te)
        assert np.array_equal(pred2, pred) and np.array_equal(probs2, probs)


@pytest.mark.gpu
@pytest.mark.usefixtures('both_gemm_precisions')
@pytest.mark.parametrize('labels,n_epochs,max_down', [('random', 40, 3), ('random', 40, 2), ('planted', 45, 2)])
def test_fit_matches_the_oracle_loop_at_cmu_shape(labels, n_epochs, max_down):
    """CMU-shape graph, [64, 64] highway, dropout 0.  Random labels: the dev loss turns after a few epochs, training stops
    early; planted labels (a linear signal smoothed over the graph): it falls throughout."""
    from geographconv_amd import synth
    from geographconv_amd.gcnmodel import GraphConv
    from geographconv_amd.nn import layers as L
    A, X, Y, (tr, dev, te), C = synth.make_graph('cmu')
    if labels == 'planted':
        rng = np.random.RandomState(11)
        S = A @ (A @ (X @ rng.randn(X.shape[1], C).astype(np.float32))) + 0.35 * rng.randn(X.shape[0], C)
        Y = S.argmax(1).astype(np.int32)
    hid = [64, 64]
    params = O.random_params(X.shape[1], hid, C, True, seed=5)
    r = _oracle_fit(params, X, A, Y, tr, dev, hid, True, 0.0, None, 0.0, n_epochs, max_down, True)
    clf = GraphConv(X.shape[1], C, hid, 0.0, 0.0, highway=True)
    clf.build_model(A, seed=77)
    L.set_all_param_values(clf.l_out, params)
    clf.fit(X, A, Y, tr, dev, n_epochs=n_epochs, max_down=max_down, verbose=False)
    _check_against_oracle(clf, r, params, n_epochs, max_down)
    if labels == 'random':
        assert r['stopped_early'] and 0 < r['best_epoch'] < r['stop_epoch'] == len(clf.fit_history) - 1
    pred, probs = clf.predict(X, A, te)
    rp, rows = O.f_val(r['trail'][clf.best_epoch], X, A, te, hid, True)
    tol = 1e-4
    assert np.abs(probs - rows).max() <= tol
    top2 = np.sort(rows, axis=1)[:, -2:]
    clear = (top2[:, 1] - top2[:, 0]) > 2 * tol
    assert clear.mean() > 0.5 and np.array_equal(pred[clear], rp[clear])


@pytest.mark.gpu
def test_fit_matches_the_oracle_loop_on_the_default_split_bf16_path():
    """A whole fit on the kernels the TwitterUS step runs, at the library's DEFAULT thresholds (no test seam): 33,001 nodes (above
    x3_rows_kernel's 32,768 rows; a ragged last row tile), [300, 300] highway, default precision 'bf16x3' -- every H . W, dZ . W^T and
    H^T . dZ of every epoch is a split-bf16 product -- against oracle.fit (the restatement of gcnmodel.py:421-449).  Labels: a linear
    signal smoothed over the graph plus noise, chosen so that the dev loss falls, wobbles and finally rises: more than 20 epochs run
    and early stopping is REACHED (max_down = 8: epoch 53 in the oracle's run)."""
    from geographconv_amd import ops, synth
    from geographconv_amd.gcnmodel import GraphConv
    from geographconv_amd.nn import layers as L
    assert ops.GEMM_PRECISION == 'bf16x3' and 'GEOGCN_X3_ROWS_MIN_M' not in os.environ
    N, C, hid, n_epochs, max_down = 33001, 16, [300, 300], 70, 8
    A = synth.powerlaw_ahat(N, N * 12, seed=4)
    X = synth.bow_x(N, 3000, 40, seed=5)
    tr, dev, te = synth.split_indices(N)
    rng = np.random.RandomState(11)
    S = A @ (A @ (X @ rng.randn(X.shape[1], C).astype(np.float32)))
    Y = (S / S.std() + 2.0 * rng.randn(N, C)).argmax(1).astype(np.int32)
    params = O.random_params(X.shape[1], hid, C, True, seed=5)
    lib = _ffi_lib()
    from geographconv_amd import _ffi
    # (the shapes of this model really are taken by the split kernel: its workspace is not the staged kernel's)
    assert lib.geogcn_gemm_dual_workspace_bytes(0, N, 300, 300, 300, _ffi.GEMM_BF16X3) > 0
    r = _oracle_fit(params, X, A, Y, tr, dev, hid, True, 0.0, None, 0.0, n_epochs, max_down, True)
    assert r['stopped_early'] and r['stop_epoch'] >= 20, (r['stop_epoch'], r['best_epoch'])
    clf = GraphConv(X.shape[1], C, hid, 0.0, 0.0, highway=True)
    clf.build_model(A, seed=77)
    L.set_all_param_values(clf.l_out, params)
    clf.fit(X, A, Y, tr, dev, n_epochs=n_epochs, max_down=max_down, verbose=False)
    assert len(clf.fit_history) >= 20 and len(clf.fit_history) < n_epochs
    _check_against_oracle(clf, r, params, n_epochs, max_down)
    pred, probs = clf.predict(X, A, te)
    rp, rows = O.f_val(r['trail'][clf.best_epoch], X, A, te, hid, True)
    tol = 1e-4
    assert np.abs(probs - rows).max() <= tol
    top2 = np.sort(rows, axis=1)[:, -2:]
    clear = (top2[:, 1] - top2[:, 0]) > 2 * tol
    assert clear.mean() > 0.5 and np.array_equal(pred[clear], rp[clear])
    # ... and the exact-fp32 fit of the same model takes the same decisions but is another computation (bits differ)
    clf32 = GraphConv(X.shape[1], C, hid, 0.0, 0.0, highway=True, gemm_precision='f32')
    clf32.build_model(A, seed=77)
    L.set_all_param_values(clf32.l_out, params)
    out32 = clf32.f_train(X, Y[tr], Y[dev], A, tr, dev)
    clfx = GraphConv(X.shape[1], C, hid, 0.0, 0.0, highway=True)
    clfx.build_model(A, seed=77)
    L.set_all_param_values(clfx.l_out, params)
    outx = clfx.f_train(X, Y[tr], Y[dev], A, tr, dev)
    assert not np.array_equal(np.asarray(outx[4]), np.asarray(out32[4]))
    assert np.abs(np.asarray(outx[4]) - np.asarray(out32[4])).max() <= 2e-6


def _ffi_lib():
    from geographconv_amd import _ffi
    return _ffi.lib()


@pytest.mark.gpu
def test_fit_log_lines_follow_the_reference_format(caplog):
    """gcnmodel.py:444: one INFO line per epoch, `best val acc` and `maxdown` as the rule sees them; :446 on stopping."""
    import logging
    from tests.helpers import make_clf
    z, A, X, params, cfg = load_case('tiny_plain_reg')
    clf = make_clf(dict(cfg, p=0.0), params)
    with caplog.at_level(logging.INFO):
        clf.fit(X, A, z['Y'], z['tr'], z['dev'], n_epochs=40, max_down=2, verbose=True)
    lines = [rec.getMessage() for rec in caplog.records]
    ep = [l for l in lines if l.startswith('epoch ')]
    assert len(ep) == len(clf.fit_history) == 6
    for e, (l, h) in enumerate(zip(ep, clf.fit_history)):
        assert l == 'epoch {} train loss {:.2f} train acc {:.2f} val loss {:.2f} val acc {:.2f} best val acc {:.2f} maxdown {}'.format(
            e, h[0], h[1], h[2], h[3], clf.fit_history[0][3], h[4])
    assert lines[0].startswith('training for 40 epochs')
    assert lines[-1] == 'validation results went down. early stopping ...'
