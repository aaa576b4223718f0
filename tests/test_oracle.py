"""The oracle checked against itself (fp64 finite differences), analytic known answers and an
independent torch-CPU autograd implementation.  The reference holds no golden vectors for this
path, so this is what "pinning" is available -- see oracle/gcn_oracle.py header."""
import numpy as np
import pytest
import scipy.sparse as sps

from geographconv_amd import synth
from oracle import gcn_oracle as O


def _tiny(N=40, V=30, C=5, hid=(8, 8, 8), highway=True, seed=0, dtype=np.float64):
    A, X, Y = synth.small_graph(N, 3.0, V, 6, C, seed=seed, empty_rows=2)
    params = O.random_params(V, list(hid), C, highway, seed=seed + 5, dtype=dtype, scale=0.6)
    tr = np.arange(0, N // 2, dtype=np.int32)
    dev = np.arange(N // 2, 3 * N // 4, dtype=np.int32)
    rng = np.random.RandomState(seed + 9)
    mask = (rng.rand(N, hid[0]) < 0.5).astype(dtype)
    return A, X, Y, params, tr, dev, mask


def _loss(params, A, X, Y, tr, hid, highway, p, mask, reg):
    c = O.forward(params, X, A, hid, highway, p, mask, deterministic=False, dtype=np.float64)
    l, _, _ = O.metrics(c['P'], tr, Y[tr])
    return l + O.reg_penalty(params, hid, highway, reg)


@pytest.mark.parametrize("highway,p,reg", [(True, 0.5, 0.0), (False, 0.0, 0.0), (True, 0.0, 1e-3)])
def test_backward_matches_fp64_finite_differences(highway, p, reg):
    hid = [8, 8, 8] if highway else [8, 6, 7]
    A, X, Y, params, tr, dev, mask = _tiny(hid=hid, highway=highway)
    c = O.forward(params, X, A, hid, highway, p, mask, deterministic=False, dtype=np.float64)
    grads = O.backward(params, c, X, A, tr, Y[tr], hid, highway, reg, dtype=np.float64)
    rng = np.random.RandomState(1)
    eps = 1e-6
    for pi, (p_arr, g) in enumerate(zip(params, grads)):
        assert g.shape == p_arr.shape
        for _ in range(6):
            idx = tuple(rng.randint(0, s) for s in p_arr.shape)
            if pi == 0 and X[:, idx[0]].nnz == 0:
                continue
            old = p_arr[idx]
            p_arr[idx] = old + eps
            lp = _loss(params, A, X, Y, tr, hid, highway, p, mask, reg)
            p_arr[idx] = old - eps
            lm = _loss(params, A, X, Y, tr, hid, highway, p, mask, reg)
            p_arr[idx] = old
            fd = (lp - lm) / (2 * eps)
            assert abs(fd - g[idx]) <= 1e-6 * max(1.0, abs(fd)) + 1e-8, (pi, idx, fd, g[idx])


@pytest.mark.parametrize("highway", [True, False])
def test_against_torch_autograd(highway):
    torch = pytest.importorskip("torch")
    hid = [8, 8, 8] if highway else [8, 6, 7]
    A, X, Y, params, tr, dev, mask = _tiny(hid=hid, highway=highway)
    p = 0.5
    c = O.forward(params, X, A, hid, highway, p, mask, deterministic=False, dtype=np.float64)
    grads = O.backward(params, c, X, A, tr, Y[tr], hid, highway, 0.0, dtype=np.float64)
    tp = [torch.tensor(q, dtype=torch.float64, requires_grad=True) for q in params]
    At = torch.tensor(A.toarray(), dtype=torch.float64)
    Xt = torch.tensor(X.toarray(), dtype=torch.float64)
    H = torch.tanh(Xt @ tp[0] + tp[1]) * torch.tensor(mask) / (1 - p)
    k = 2
    for _ in range(len(hid) - 1):
        if highway:
            Wt, bt, Wh, bh = tp[k:k + 4]
            k += 4
            Hc = torch.tanh(At @ (H @ Wh) + bh)
            T = torch.sigmoid(H @ Wt + bt)
            H = T * Hc + (1 - T) * H
        else:
            Wi, bi = tp[k:k + 2]
            k += 2
            H = torch.tanh(At @ (H @ Wi) + bi)
    P = torch.softmax(At @ (H @ tp[k]) + tp[k + 1], dim=1)
    assert np.allclose(P.detach().numpy(), c['P'], rtol=1e-10, atol=1e-12)
    loss = -torch.log(P[torch.tensor(tr, dtype=torch.long), torch.tensor(Y[tr], dtype=torch.long)]).mean()
    loss.backward()
    for q, g in zip(tp, grads):
        assert np.allclose(q.grad.numpy(), g, rtol=1e-8, atol=1e-12)


def test_identity_graph_is_dense_mlp():
    A, X, Y, params, tr, dev, mask = _tiny(hid=[8, 8], highway=False)
    I = sps.identity(X.shape[0], dtype=np.float64, format='csr')
    c = O.forward(params, X, I, [8, 8], False, dtype=np.float64)
    Xd = X.toarray().astype(np.float64)
    H = np.tanh(Xd @ params[0] + params[1])
    H = np.tanh(H @ params[2] + params[3])
    P = O.softmax_rows(H @ params[4] + params[5])
    assert np.allclose(c['P'], P, rtol=1e-12, atol=1e-14)
    assert np.allclose(c['P'].sum(axis=1), 1.0)


def test_gate_closed_form_and_uniform_ce():
    # Wt = 0, bt = -4  =>  T = sigma(-4) everywhere (highway_dense default gcnmodel.py:274)
    A, X, Y, params, tr, dev, mask = _tiny(hid=[8, 8], highway=True)
    params[2][:] = 0
    params[3][:] = -4.0
    c = O.forward(params, X, A, [8, 8], True, dtype=np.float64)
    assert np.allclose(c['blocks'][0]['T'], 0.01798620996209156)
    # zero output layer => uniform probs => CE = log C
    params[-2][:] = 0
    params[-1][:] = 0
    c = O.forward(params, X, A, [8, 8], True, dtype=np.float64)
    l, _, _ = O.metrics(c['P'], tr, Y[tr])
    assert abs(l - np.log(5)) < 1e-12


def test_lasagne_adam_three_steps_hand_computed():
    # lasagne.updates.adam, SURVEY.md A.4: a_t = lr*sqrt(1-b2^t)/(1-b1^t); p -= a_t*m/(sqrt(v)+eps)
    p = [np.array([1.0, -2.0])]
    st = O.AdamState(p)
    lr, b1, b2, eps = 2e-3, 0.9, 0.999, 1e-8
    m = np.zeros(2)
    v = np.zeros(2)
    ref = p[0].copy()
    for t in range(1, 4):
        g = 2 * ref                      # d/dp of p^2
        m = b1 * m + (1 - b1) * g
        v = b2 * v + (1 - b2) * g * g
        a = lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t)
        ref = ref - a * m / (np.sqrt(v) + eps)
        p = O.adam_step(p, [2 * p[0]], st, lr, b1, b2, eps)
        assert np.allclose(p[0], ref, rtol=1e-14)
    # first step moves each coordinate by ~lr regardless of gradient scale
    assert st.t == 3


def test_fp32_vs_fp64_envelope_defines_tolerance():
    """Derives the tolerance the GPU parity tests state: fp32 oracle vs fp64 oracle."""
    s = synth.CMU
    A, X, Y = synth.small_graph(2000, 7.0, 500, 30, 20, seed=3)
    hid = [300, 300, 300]
    params = O.random_params(500, hid, 20, True, seed=1)
    c32 = O.forward(params, X, A, hid, True, dtype=np.float32)
    c64 = O.forward(params, X, A, hid, True, dtype=np.float64)
    err = np.abs(c32['logits'] - c64['logits']).max()
    assert err < 2e-5, err
    assert np.abs(c32['P'] - c64['P']).max() < 1e-6


def test_f_train_dev_metrics_use_dropout_pass():
    A, X, Y, params, tr, dev, mask = _tiny(hid=[8, 8, 8], highway=True, dtype=np.float32)
    st = O.AdamState(params)
    newp, outs, grads = O.f_train(params, st, X, Y[tr], Y[dev], A, tr, dev, [8, 8, 8], True, 0.5,
                                  mask.astype(np.float32))
    c = O.forward(params, X, A, [8, 8, 8], True, 0.5, mask, deterministic=False)
    l_dev, a_dev, _ = O.metrics(c['P'], dev, Y[dev])
    assert outs[2] == l_dev and outs[3] == a_dev
    assert outs[4].shape == (X.shape[0], 5)
    pred, probs = O.f_val(newp, X, A, dev, [8, 8, 8], True)
    assert pred.dtype == np.int64 and probs.shape == (len(dev), 5)


def test_multithreaded_cpu_leg_equals_the_oracle_products():
    """oracle/cpu_mt.c (bench.py's `cpu_baseline_mt`): OpenMP over the rows, same per-row order as the single-threaded
    loop -- equal to scipy's product up to fma contraction; patched() swaps it into the oracle for one timed step."""
    import scipy.sparse as sps
    from oracle import cpu_mt
    cpu_mt.build()
    assert cpu_mt.threads() >= 1
    A = sps.random(3000, 2000, density=0.01, format='csr', dtype=np.float32, random_state=0)
    B = np.random.RandomState(0).randn(2000, 129).astype(np.float32)
    ref = O.spmm(A, B)
    got = cpu_mt.spmm(A, B)
    assert np.abs(got - ref).max() <= 4e-6 * np.asarray(abs(A) @ np.abs(B)).max()
    G = np.random.RandomState(1).randn(3000, 7).astype(np.float32)
    with cpu_mt.patched(O, {id(A): sps.csr_matrix(A.T)}):
        r = O.spmm_t(A, G)
        r64 = O.spmm_t(A, G.astype(np.float64))                 # fp64 calls (gradient checks) stay on scipy
    assert np.abs(r - A.T @ G).max() <= 1e-5 and r64.dtype == np.float64
    assert O.spmm_t(A, G).dtype == np.float32 and O.spmm.__name__ == 'spmm'      # restored


def test_multithreaded_cpu_step_equals_the_oracle_step():
    """oracle/cpu_mt.py::f_train (bench.py's `cpu_baseline_mt`: every pass of the step on all cores) against oracle.f_train:
    three steps with and without dropout -- losses, hit counts, probabilities, every gradient, the Adam update."""
    import scipy.sparse as sps
    from oracle import cpu_mt
    cpu_mt.build()
    A, X, Y = synth.small_graph(1500, 6.0, 200, 12, 9, seed=3)
    hid = [24, 24, 24]
    params = O.random_params(200, hid, 9, True, seed=4)
    tr, dev = np.arange(0, 900), np.arange(900, 1200)
    mask = (np.random.RandomState(5).rand(1500, 24) < 0.5).astype(np.float32)
    At, Xt = sps.csr_matrix(A.T), sps.csr_matrix(X.T)
    for p in (0.5, 0.0):
        st1, st2 = O.AdamState(params), O.AdamState(params)
        cur1, cur2 = [q.copy() for q in params], [q.copy() for q in params]
        for step in range(3):
            cur1, o1, g1 = O.f_train(cur1, st1, X, Y[tr], Y[dev], A, tr, dev, hid, True, p, mask)
            cur2, o2, g2 = cpu_mt.f_train(O, cur2, st2, X, Xt, Y[tr], Y[dev], A, At, tr, dev, hid, p, mask)
            assert abs(o1[0] - o2[0]) <= 2e-6 * abs(o1[0]) and abs(o1[2] - o2[2]) <= 2e-6 * abs(o1[2])
            assert o1[1] == o2[1] and o1[3] == o2[3]
            assert np.abs(o1[4] - o2[4]).max() <= 1e-6
            for a, b in zip(g1, g2):
                assert np.abs(a - b).max() <= 1e-5 * np.abs(a).max() + 1e-10
            for a, b in zip(cur1, cur2):
                assert np.abs(a - b).max() <= 2e-3 * 0.02 + 1e-7


def test_bf16_aware_mode_rounds_where_the_bf16_configuration_rounds():
    """`gemm_operands='bf16'` (BASELINE configs[4]; no reference counterpart): RNE rounding incl. ties, idempotence, a dense
    product whose operands are already bf16 is unchanged, the mode stays within a bf16-class distance of the fp32 forward,
    and the first layer (X.W0: not an H.W product) is untouched."""
    # ties to even: 1 + 2^-8 lies exactly between 1 and 1 + 2^-7 -> 1 (even mantissa); 1 + 3 * 2^-8 -> 1 + 2^-6
    x = np.array([1.0 + 2.0 ** -8, 1.0 + 3 * 2.0 ** -8, -(1.0 + 2.0 ** -8), 1.0 + 2.0 ** -8 + 2.0 ** -20, 0.0, 3.0e38],
                 dtype=np.float32)
    r = O.bf16_round(x)
    assert r[0] == 1.0 and r[1] == np.float32(1.0 + 2.0 ** -6) and r[2] == -1.0 and r[3] == np.float32(1.0 + 2.0 ** -7)
    assert r[4] == 0.0 and np.isfinite(r[5])
    y = np.random.RandomState(0).randn(1000).astype(np.float32)
    assert np.array_equal(O.bf16_round(O.bf16_round(y)), O.bf16_round(y))
    assert np.all(np.abs(O.bf16_round(y) - y) <= np.abs(y) * 2.0 ** -8)
    H, W = O.bf16_round(np.random.RandomState(1).randn(40, 30).astype(np.float32)), O.bf16_round(
        np.random.RandomState(2).randn(30, 20).astype(np.float32))
    assert np.allclose(O.dense_product(H, W, 'bf16'), H.astype(np.float64) @ W.astype(np.float64), rtol=1e-6, atol=1e-6)
    A, X, Y = synth.small_graph(300, 5.0, 60, 10, 6, seed=3)
    hid = [24, 24, 24]
    params = O.random_params(60, hid, 6, True, seed=4, scale=0.4)
    c32 = O.forward(params, X, A, hid, True)
    cb = O.forward(params, X, A, hid, True, gemm_operands='bf16')
    assert np.array_equal(c32['H0'], cb['H0'])
    d = np.abs(c32['P'] - cb['P']).max()
    assert 0 < d < 2e-2
    assert np.array_equal(cb['blocks'][0]['Z'], O.bf16_round(cb['blocks'][0]['Z']))          # Z is STORED as bf16
    with pytest.raises(ValueError):
        O.forward(params, X, A, hid, True, dtype=np.float64, gemm_operands='bf16')
