"""Randomised (seeded, reproducible) sweeps over the C-ABI entry points: odd sizes, padded pitches, empty rows / empty
matrices, rows around the long-row threshold, every epilogue -- each case against an fp64 NumPy product with an envelope
scaled by the magnitudes that enter the sum (|A|.|B|), i.e. the tolerance is that of fp32 summation in ANY order:
   |got - ref| <= 4e-6 * (|A| . |B| [+ |bias|]) * act_lipschitz + 1e-6
The fixed-shape tests in test_kernels_gpu.py pin the interesting corners; this file is the net under them."""
import numpy as np
import pytest
import scipy.sparse as sps

import os

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

# GEOGCN_FUZZ_SCALE=10 multiplies the number of seeds of every sweep (a longer soak; the default runs in ~10 s)
_SCALE = max(1, int(os.environ.get('GEOGCN_FUZZ_SCALE', '1')))


def _seeds(n):
    return range(n * _SCALE)


@pytest.fixture(scope="module")
def dev():
    from geographconv_amd import ops
    ops.require_gpu()
    return torch.device("cuda:0")


def _act(x, act):
    if act == 1:
        return np.tanh(x)
    if act == 2:
        return 1.0 / (1.0 + np.exp(-x))
    return x


def _dmat(ops, dev, a, rng, pad=True):
    """Device matrix with a random legal pitch (>= pad4(F), multiple of 4), pads zero."""
    n, F = a.shape
    ld = ops.pad4(F) + (4 * rng.randint(0, 3) if pad else 0)
    m = ops.DMat.empty(n, F, dev, ld=max(ld, 4))
    m.t.zero_()
    if n and F:
        m.t[:, :F].copy_(torch.from_numpy(np.ascontiguousarray(a)))
    return m


def _csr(rng, n_rows, n_cols, mean, long_every=0, long_nnz=0):
    rows, cols = [], []
    for r in range(n_rows):
        k = 0 if rng.rand() < 0.15 else min(n_cols, rng.poisson(mean))
        if long_every and r % long_every == 1:
            k = min(n_cols, long_nnz)
        if k:
            rows += [r] * k
            cols += list(rng.choice(n_cols, size=k, replace=False))
    vals = rng.randn(len(rows)).astype(np.float32)
    m = sps.csr_matrix((vals, (np.asarray(rows, dtype=np.int64), np.asarray(cols, dtype=np.int64))), shape=(n_rows, n_cols),
                       dtype=np.float32)
    m.sort_indices()
    return m


def _close(got, ref, mag, what, slack=4e-6):
    err = np.abs(got.astype(np.float64) - ref)
    bound = slack * mag + 1e-6
    assert np.all(err <= bound), (what, float(err.max()), float((err - bound).max()))


@pytest.mark.parametrize("seed", _seeds(6))
def test_gemm_random_shapes(dev, seed):
    """geogcn_gemm_f32 in all four layouts that the path uses, with bias / activation / accumulate."""
    from geographconv_amd import ops
    rng = np.random.RandomState(1000 + seed)
    for case in range(14):
        M, N, K = (int(rng.choice([1, 3, 17, 64, 100, 129, 255, 300, 321, 700])) for _ in range(3))
        transA, transB = bool(rng.randint(2)), bool(rng.randint(2))
        if transA:
            transB = False                                   # (A^T . B^T is never formed on the path)
        act = int(rng.choice([0, 1, 2])) if not transA else 0
        use_bias = bool(rng.randint(2)) and not transA
        accumulate = bool(rng.randint(2))
        A = rng.randn(K, M).astype(np.float32) if transA else rng.randn(M, K).astype(np.float32)
        B = rng.randn(N, K).astype(np.float32) if transB else rng.randn(K, N).astype(np.float32)
        bias = rng.randn(N).astype(np.float32) if use_bias else None
        C0 = rng.randn(M, N).astype(np.float32)
        a64 = (A.T if transA else A).astype(np.float64)
        b64 = (B.T if transB else B).astype(np.float64)
        pre = a64 @ b64 + (bias.astype(np.float64) if use_bias else 0.0)
        ref = _act(pre, act) + (C0 if accumulate else 0.0)
        mag = np.abs(a64) @ np.abs(b64) + (np.abs(bias) if use_bias else 0.0) + (np.abs(C0) if accumulate else 0.0)
        dA, dB = _dmat(ops, dev, A, rng), _dmat(ops, dev, B, rng)
        out = _dmat(ops, dev, C0, rng)
        bt = None
        if use_bias:
            bt = torch.zeros(ops.pad4(N), device=dev)
            bt[:N] = torch.from_numpy(bias)
        got = ops.gemm(dA, dB, out=out, transA=transA, transB=transB, bias=bt, act=act, accumulate=accumulate)
        _close(got.numpy(), ref, mag, ('gemm', seed, case, M, N, K, transA, transB, act, use_bias, accumulate))
        assert torch.all(got.t[:, N:ops.pad4(N)] == 0), ('pad columns', seed, case)


@pytest.mark.parametrize("seed", _seeds(4))
def test_gemm_dual_and_kcat_random_shapes(dev, seed):
    """The highway block's fused launches: two weights on one input (plain and transposed input), two products into one
    accumulator."""
    from geographconv_amd import ops
    rng = np.random.RandomState(2000 + seed)
    for case in range(8):
        M = int(rng.choice([5, 130, 257, 1000, 2311]))
        K = int(rng.choice([3, 64, 129, 300]))
        N0, N1 = (int(rng.choice([1, 40, 129, 300])) for _ in range(2))
        H = rng.randn(M, K).astype(np.float32)
        W0, W1 = rng.randn(K, N0).astype(np.float32) * 0.2, rng.randn(K, N1).astype(np.float32) * 0.2
        b1 = rng.randn(N1).astype(np.float32)
        dH, dW0, dW1 = _dmat(ops, dev, H, rng), _dmat(ops, dev, W0, rng), _dmat(ops, dev, W1, rng)
        bt = torch.zeros(ops.pad4(N1), device=dev)
        bt[:N1] = torch.from_numpy(b1)
        z, t = ops.gemm_dual(dH, dW0, dW1, bias1=bt, act1=ops.ACT_SIGMOID)
        h64 = H.astype(np.float64)
        _close(z.numpy(), h64 @ W0, np.abs(h64) @ np.abs(W0), ('dual z', seed, case))
        _close(t.numpy(), _act(h64 @ W1 + b1, 2), np.abs(h64) @ np.abs(W1) + np.abs(b1), ('dual t', seed, case))
        # transposed input: (K x N0, K x N1) = H^T . (G0, G1) with M the reduction
        G0, G1 = rng.randn(M, N0).astype(np.float32), rng.randn(M, N1).astype(np.float32)
        g0, g1 = ops.gemm_dual(dH, _dmat(ops, dev, G0, rng), _dmat(ops, dev, G1, rng), transA=True)
        _close(g0.numpy(), h64.T @ G0, np.abs(h64.T) @ np.abs(G0), ('dual tn 0', seed, case))
        _close(g1.numpy(), h64.T @ G1, np.abs(h64.T) @ np.abs(G1), ('dual tn 1', seed, case))
        # k-concatenated: out (M x K) = G0 . W0^T + G1 . W1^T [+ out]
        C0 = rng.randn(M, K).astype(np.float32)
        out = _dmat(ops, dev, C0, rng)
        acc = bool(rng.randint(2))
        got = ops.gemm_kcat(_dmat(ops, dev, G0, rng), dW0, _dmat(ops, dev, G1, rng), dW1, out=out, transB=True, accumulate=acc)
        ref = G0.astype(np.float64) @ W0.T + G1.astype(np.float64) @ W1.T + (C0 if acc else 0.0)
        mag = np.abs(G0) @ np.abs(W0.T) + np.abs(G1) @ np.abs(W1.T) + (np.abs(C0) if acc else 0.0)
        _close(got.numpy(), ref, mag, ('kcat', seed, case, acc))
        # ... with the highway block's carry gradient G * (1 - T) formed in the epilogue (round 4), for two products and for one
        # (every precision), and on its own
        Gc, Tc = rng.randn(M, K).astype(np.float32), rng.rand(M, K).astype(np.float32)
        gc = ops.GateCarry(_dmat(ops, dev, Gc, rng), _dmat(ops, dev, Tc, rng))
        carry = Gc.astype(np.float64) * (1.0 - Tc.astype(np.float64))
        _close(gc.dense().numpy(), carry, np.abs(carry), ('gate carry', seed, case))
        got = ops.gemm_kcat(_dmat(ops, dev, G0, rng), dW0, _dmat(ops, dev, G1, rng), dW1, transB=True, gate_carry=gc)
        ref = G0.astype(np.float64) @ W0.T + G1.astype(np.float64) @ W1.T + carry
        mag = np.abs(G0) @ np.abs(W0.T) + np.abs(G1) @ np.abs(W1.T) + np.abs(carry)
        _close(got.numpy(), ref, mag, ('kcat gated', seed, case))
        for prec, tol in (('f32', None), ('bf16x3', None), ('bf16', 2e-2)):
            got = ops.gemm(_dmat(ops, dev, G0, rng), dW0, transB=True, precision=prec, gate_carry=gc)
            ref1 = G0.astype(np.float64) @ W0.T + carry
            if tol is None:
                _close(got.numpy(), ref1, np.abs(G0) @ np.abs(W0.T) + np.abs(carry), ('gemm gated', prec, seed, case))
            else:
                assert np.abs(got.numpy() - ref1).max() <= tol * (np.abs(G0) @ np.abs(W0.T) + np.abs(carry)).max() + 1e-6, (prec, seed, case)
        # the bf16 configuration's forward pair in one launch (round 4)
        z16, t32 = ops.gemm_dual_bf16(dH, dW0, dW1, bias1=bt, act1=ops.ACT_SIGMOID)
        zr, tr_ = h64 @ W0, _act(h64 @ W1 + b1, 2)
        assert np.abs(z16.numpy() - zr).max() <= 2e-2 * (np.abs(h64) @ np.abs(W0)).max() + 1e-6, ('dual bf16 z', seed, case)
        assert np.abs(t32.numpy() - tr_).max() <= 2e-2 * max(1.0, (np.abs(h64) @ np.abs(W1)).max()), ('dual bf16 t', seed, case)


@pytest.mark.parametrize("seed", _seeds(6))
def test_spmm_random_structures(dev, seed):
    """geogcn_spmm_csr_f32 / _acc_f32 / _highway_f32 on random structures: empty rows, empty matrices, rows around the
    long-row threshold (forced low so that the chunked path runs), random widths and pitches."""
    from geographconv_amd import ops
    rng = np.random.RandomState(3000 + seed)
    for case in range(10):
        n_rows = int(rng.choice([1, 2, 33, 257, 900]))
        n_cols = int(rng.choice([1, 7, 300, 1200]))
        F = int(rng.choice([1, 4, 37, 64, 129, 300, 333]))
        long_nnz = int(rng.choice([48, 256]))
        A = _csr(rng, n_rows, n_cols, mean=rng.choice([0, 3, 20]), long_every=int(rng.choice([0, 9])), long_nnz=long_nnz + 70)
        B = rng.randn(n_cols, F).astype(np.float32)
        act = int(rng.choice([0, 1]))
        bias = rng.randn(F).astype(np.float32) if rng.randint(2) else None
        dA = ops.CSR(A, dev, long_row_nnz=long_nnz, chunk_nnz=int(rng.choice([16, 128])))
        dB = _dmat(ops, dev, B, rng)
        bt = None
        if bias is not None:
            bt = torch.zeros(ops.pad4(F), device=dev)
            bt[:F] = torch.from_numpy(bias)
        a64 = A.astype(np.float64)
        pre = np.asarray(a64 @ B.astype(np.float64)) + (bias if bias is not None else 0.0)
        mag = np.asarray(abs(a64) @ np.abs(B).astype(np.float64)) + (np.abs(bias) if bias is not None else 0.0)
        got = ops.spmm(dA, dB, bias=bt, act=act)
        _close(got.numpy(), _act(pre, act), mag, ('spmm', seed, case, n_rows, n_cols, F, long_nnz))
        # accumulate form: out = act(out + A . B + b)
        C0 = rng.randn(n_rows, F).astype(np.float32)
        out = _dmat(ops, dev, C0, rng, pad=False)
        got = ops.spmm(dA, dB, out=out, bias=bt, act=act, accumulate=True)
        _close(got.numpy(), _act(pre + C0, act), mag + np.abs(C0), ('spmm acc', seed, case))
        # bf16 gathered operand (the bf16 configuration): exact product with the ROUNDED operand, fp32 accumulation
        Bq = torch.from_numpy(B).to(torch.bfloat16).float().numpy()
        hB = ops.cast_bf16(dB)
        assert np.array_equal(hB.numpy(), Bq), ('cast_bf16', seed, case)
        pre_q = np.asarray(a64 @ Bq.astype(np.float64)) + (bias if bias is not None else 0.0)
        mag_q = np.asarray(abs(a64) @ np.abs(Bq).astype(np.float64)) + (np.abs(bias) if bias is not None else 0.0)
        got = ops.spmm(dA, hB, bias=bt, act=act)
        _close(got.numpy(), _act(pre_q, act), mag_q, ('spmm bf16 operand', seed, case, n_rows, n_cols, F))
        # highway epilogue: Hout = T * tanh(A . B + b) + (1 - T) * H
        if F % 4 == 0 or True:
            T = rng.rand(n_rows, F).astype(np.float32)
            Hm = rng.randn(n_rows, F).astype(np.float32)
            ld = ops.pad4(F)
            dT, dHm = ops.DMat.empty(n_rows, F, dev, ld=ld), ops.DMat.empty(n_rows, F, dev, ld=ld)
            for d, a in ((dT, T), (dHm, Hm)):
                d.t.zero_()
                if n_rows:
                    d.t[:, :F].copy_(torch.from_numpy(a))
            hc, hout = ops.spmm_highway(dA, dB, bt, dT, dHm)
            ref_hc = np.tanh(pre)
            _close(hc.numpy(), ref_hc, mag, ('highway hc', seed, case))
            _close(hout.numpy(), T * ref_hc + (1.0 - T) * Hm, mag + np.abs(Hm), ('highway out', seed, case))


@pytest.mark.parametrize("seed", _seeds(4))
def test_bow_products_random(dev, seed, monkeypatch):
    """X . W0 with the hot rows of W0 in LDS and X^T . G through the document-blocked kernel, random bag-of-words
    structures (Zipfian columns), against fp64."""
    from geographconv_amd import ops, synth, tuning
    monkeypatch.setattr(tuning, 'XT_MIN_NNZ', 0)
    monkeypatch.setattr(tuning, 'HOT_MIN_NNZ', 0)
    rng = np.random.RandomState(4000 + seed)
    for case in range(4):
        n_docs = int(rng.choice([300, 2100, 9000]))
        n_words = int(rng.choice([40, 333, 1500]))
        F = int(rng.choice([5, 64, 129, 300, 340]))
        X = sps.csr_matrix(synth.bow_x(n_docs, n_words, int(rng.choice([3, 12, 30])), seed=seed * 10 + case))
        X.data[::5] *= -1.0
        W = rng.randn(n_words, F).astype(np.float32)
        G = rng.randn(n_docs, F).astype(np.float32)
        b = rng.randn(F).astype(np.float32)
        x = ops.SparseOperand.from_scipy(X, dev, dense_head=bool(rng.randint(2)))
        bt = torch.zeros(ops.pad4(F), device=dev)
        bt[:F] = torch.from_numpy(b)
        x64 = X.astype(np.float64)
        got = ops.spmm_x(x, _dmat(ops, dev, W, rng), bias=bt, act=1)
        _close(got.numpy(), np.tanh(np.asarray(x64 @ W.astype(np.float64)) + b), np.asarray(abs(x64) @ np.abs(W)) + np.abs(b),
               ('X.W0', seed, case, n_docs, n_words, F))
        got = ops.spmm_t(x, _dmat(ops, dev, G, rng))
        _close(got.numpy(), np.asarray(x64.T @ G.astype(np.float64)), np.asarray(abs(x64).T @ np.abs(G)),
               ('X^T.G', seed, case, n_docs, n_words, F))


@pytest.mark.parametrize("seed", _seeds(3))
def test_softmax_ce_random(dev, seed):
    """Row softmax + argmax, cross-entropy sums / hit counts on random index sets (with repeats), CE gradient."""
    from geographconv_amd import ops
    rng = np.random.RandomState(5000 + seed)
    for case in range(6):
        N = int(rng.choice([1, 50, 1000]))
        C = int(rng.choice([2, 3, 129, 256, 700]))
        L = (rng.randn(N, C) * rng.choice([0.1, 3.0])).astype(np.float32)
        dL = _dmat(ops, dev, L, rng)
        am = torch.zeros(N, dtype=torch.int32, device=dev)
        P = ops.softmax_rows(dL, argmax=am)
        e = np.exp(L.astype(np.float64) - L.max(axis=1, keepdims=True))
        ref = e / e.sum(axis=1, keepdims=True)
        assert np.allclose(P.numpy(), ref, rtol=2e-6, atol=1e-7), ('softmax', seed, case)
        srt = np.sort(ref, axis=1)
        clear = (srt[:, -1] - srt[:, -2]) > 1e-6 if C > 1 else np.ones(N, bool)
        assert np.array_equal(am.cpu().numpy()[clear], ref.argmax(axis=1)[clear]), ('argmax', seed, case)


@pytest.mark.parametrize("seed", _seeds(3))
def test_gemm_panel_output_and_reduced_precisions_random(dev, seed):
    """The product written as feature panels (send layout of the multi-GPU repartition), fp32 and bf16 panels; and the
    bf16 / bf16x3 arithmetic against fp64 with their own envelopes (bf16: 2^-8 per operand; bf16x3: fp32 class)."""
    from geographconv_amd import ops
    rng = np.random.RandomState(6000 + seed)
    for case in range(6):
        M = int(rng.choice([7, 200, 1031]))
        K = int(rng.choice([5, 64, 300]))
        N = int(rng.choice([3, 40, 129, 300]))
        W = int(rng.choice([1, 2, 3, 8]))
        transB = bool(rng.randint(2))
        A = rng.randn(M, K).astype(np.float32)
        B = (rng.randn(N, K) if transB else rng.randn(K, N)).astype(np.float32)
        a64, b64 = A.astype(np.float64), (B.T if transB else B).astype(np.float64)
        ref, mag = a64 @ b64, np.abs(a64) @ np.abs(b64)
        dA, dB = _dmat(ops, dev, A, rng), _dmat(ops, dev, B, rng)
        for bf16 in (False, True):
            q = 8 if bf16 else 4
            wp = (-(-N // W) + q - 1) // q * q
            R = M + int(rng.randint(0, 5))
            p = ops.Panels(M, N, R, W, wp, dev, bf16=bf16)
            ops.gemm(dA, dB, out=p, transB=transB)
            t = p.t.float().view(W, R, wp)[:, :M, :].permute(1, 0, 2).reshape(M, W * wp).cpu().numpy()
            if bf16:
                # bf16 operands (2^-9 relative each) and a bf16 result (2^-9 relative)
                assert np.all(np.abs(t[:, :N] - ref) <= 2.0 ** -7 * mag + 2.0 ** -8 * np.abs(ref) + 1e-6), ('bf16 panels', seed, case)
            else:
                _close(t[:, :N], ref, mag, ('panels', seed, case, M, N, K, W, wp))
            # (columns >= N are left untouched by contract -- include/geogcn.h; the buffer starts zeroed, nothing may leak in)
            assert np.all(t[:, N:] == 0), ('panel pad columns', seed, case, bf16)
        got = ops.gemm(dA, dB, transB=transB, precision='bf16x3')
        _close(got.numpy(), ref, mag, ('bf16x3', seed, case), slack=8e-6)
        got = ops.gemm(dA, dB, transB=transB, precision='bf16')
        assert np.all(np.abs(got.numpy() - ref) <= 2.0 ** -7 * mag + 1e-6), ('bf16', seed, case)
        if K >= 64:
            G = rng.randn(M, N).astype(np.float32)
            got = ops.gemm(dA, _dmat(ops, dev, G, rng), transA=True, precision='bf16')          # dW = A^T . G, M the reduction
            assert np.all(np.abs(got.numpy() - a64.T @ G) <= 2.0 ** -7 * (np.abs(a64.T) @ np.abs(G)) + 1e-6), ('bf16 tn', seed, case)


@pytest.mark.parametrize("seed", _seeds(24))
def test_training_step_random_models(dev, seed):
    """Whole f_train steps of randomly shaped models (odd widths, highway on / off, dropout with an injected mask,
    L1+L2, index sets with gaps, graphs with empty rows) against the CPU restatement: losses, hit counts, probabilities,
    every gradient, the Adam update -- then f_val."""
    from geographconv_amd.nn import layers as L
    from geographconv_amd import synth
    from oracle import gcn_oracle as O
    from tests.helpers import make_clf
    rng = np.random.RandomState(7000 + seed)
    N = int(rng.choice([19, 64, 150, 333]))
    V = int(rng.choice([11, 40, 97]))
    C = int(rng.choice([2, 3, 7, 33]))
    highway = bool(rng.randint(2))
    depth = int(rng.choice([1, 2, 3]))
    w = int(rng.choice([4, 9, 16, 37]))
    hid = [w] * depth if highway else [int(rng.choice([4, 9, 16, 37])) for _ in range(depth)]
    p = float(rng.choice([0.0, 0.3, 0.5]))
    reg = float(rng.choice([0.0, 1e-3]))
    A, X, Y = synth.small_graph(N, 4.0, V, 6, C, seed=seed, empty_rows=int(rng.randint(0, 3)))
    params = O.random_params(V, hid, C, highway, seed=seed + 1, scale=0.5)
    perm = rng.permutation(N)
    n_tr, n_dev = max(1, int(0.5 * N)), max(1, int(0.2 * N))
    tr = np.sort(perm[:n_tr]).astype(np.int32)
    dv = np.sort(perm[n_tr:n_tr + n_dev]).astype(np.int32)
    te = np.sort(perm[n_tr + n_dev:]).astype(np.int32)
    mask = (rng.rand(N, hid[0]) < (1 - p)).astype(np.uint8) if p > 0 else np.ones((N, hid[0]), np.uint8)
    cfg = dict(N=N, V=V, C=C, hid=hid, highway=highway, p=p, reg=reg)
    clf = make_clf(cfg, [q.copy() for q in params], device=dev)
    clf.inject_dropout_mask(mask)
    st = O.AdamState(params)
    cur = [q.copy() for q in params]
    what = (seed, cfg)
    for step in range(2):
        new, outs, grads = O.f_train(cur, st, X, Y[tr], Y[dv], A, tr, dv, hid, highway, p, mask.astype(np.float32), reg)
        o = clf.f_train(X, Y[tr], Y[dv], A, tr, dv)
        assert np.allclose([float(v) for v in o[:4]], outs[:4], rtol=2e-5, atol=2e-6), (what, step, o[:4], outs[:4])
        assert np.allclose(np.asarray(o[4]), outs[4], rtol=2e-4, atol=2e-6), (what, step)
        for i, (g, r) in enumerate(zip(clf.get_grads(), grads)):
            assert np.allclose(g, r, rtol=5e-4, atol=5e-7 + 2e-5 * np.abs(r).max()), (what, step, i, np.abs(g - r).max())
        for i, (q, r) in enumerate(zip(L.get_all_param_values(clf.l_out), new)):
            assert np.allclose(q, r, rtol=1e-4, atol=5e-5), (what, step, 'param', i, np.abs(q - r).max())
        cur = new
    # continue the comparison from the DEVICE's parameters (Adam amplifies rounding differences of tiny gradients)
    cur = [q.copy() for q in L.get_all_param_values(clf.l_out)]
    pred, probs = O.f_val(cur, X, A, te, hid, highway)
    gp, gprobs = clf.predict(X, A, te)
    assert np.allclose(gprobs, probs, rtol=2e-4, atol=2e-6), what
    srt = np.sort(probs, axis=1)
    clear = (srt[:, -1] - srt[:, -2]) > 1e-5
    assert np.array_equal(gp[clear], pred[clear]), what


@pytest.mark.parametrize("seed", _seeds(4))
def test_elementwise_random(dev, seed):
    """Highway mix and its backward (with and without the fused column sums), activation backward (+ column sums,
    + dropout mask), dropout apply, column sums, pack / unpack of feature panels -- random sizes and pitches."""
    from geographconv_amd import ops
    rng = np.random.RandomState(8000 + seed)
    for case in range(8):
        n = int(rng.choice([1, 2, 63, 64, 65, 700, 4099]))
        F = int(rng.choice([1, 3, 4, 37, 128, 300, 301]))
        T = rng.rand(n, F).astype(np.float32)
        Hc = np.tanh(rng.randn(n, F)).astype(np.float32)
        H, G = rng.randn(n, F).astype(np.float32), rng.randn(n, F).astype(np.float32)
        ld = ops.pad4(F)
        mk = lambda a: _dmat(ops, dev, a, rng, pad=False)
        dT, dHc, dH, dG = mk(T), mk(Hc), mk(H), mk(G)
        out = ops.highway_fwd(dT, dHc, dH)
        assert np.allclose(out.numpy(), T * Hc + (1 - T) * H, rtol=1e-6, atol=1e-6), ('highway_fwd', seed, case)
        rS = G * T * (1 - Hc * Hc)
        rU = G * (Hc - H) * T * (1 - T)
        rC = G * (1 - T)
        for with_sums in (False, True):
            dbS = torch.full((ld,), 9.0, device=dev) if with_sums else None
            dbU = torch.full((ld,), 9.0, device=dev) if with_sums else None
            dS, dU, dC = ops.highway_bwd(dG, dT, dHc, dH, dbS=dbS, dbU=dbU)
            for got, ref, nm in ((dS, rS, 'dS'), (dU, rU, 'dU'), (dC, rC, 'dHcarry')):
                assert np.allclose(got.numpy(), ref, rtol=2e-6, atol=1e-6), (nm, seed, case, with_sums)
            if with_sums:
                for got, ref, nm in ((dbS, rS, 'dbS'), (dbU, rU, 'dbU')):
                    r64 = ref.astype(np.float64).sum(axis=0)
                    assert np.all(np.abs(got.cpu().numpy()[:F] - r64) <= 4e-6 * np.abs(ref).sum(axis=0) + 1e-6), (nm, seed, case)
        keep = (rng.rand(n, F) < 0.6).astype(np.uint8)
        dk = torch.from_numpy(keep).to(dev)
        for act, dfun in ((1, lambda y: 1 - y * y), (2, lambda y: y * (1 - y))):
            Yv = (np.tanh(H) if act == 1 else T).astype(np.float32)
            dY = mk(Yv)
            for km, scale in ((None, 1.0), (dk, 1.0 / 0.6)):
                m = 1.0 if km is None else keep * np.float32(scale)
                ref = G * m * dfun(Yv)
                got = ops.act_bwd(dG, dY, act, keep_mask=km, scale=scale)
                assert np.allclose(got.numpy(), ref, rtol=3e-6, atol=1e-6), ('act_bwd', seed, case, act)
                db = torch.full((ld,), 9.0, device=dev)
                got = ops.act_bwd_colsum(dG, dY, act, db, keep_mask=km, scale=scale)
                assert np.allclose(got.numpy(), ref, rtol=3e-6, atol=1e-6), ('act_bwd_colsum', seed, case, act)
                assert np.all(np.abs(db.cpu().numpy()[:F] - ref.astype(np.float64).sum(axis=0)) <= 4e-6 * np.abs(ref).sum(axis=0) + 1e-6)
        got = ops.dropout_apply(dH, dk, 0.4)
        assert np.allclose(got.numpy(), H * keep * np.float32(1 / 0.6), rtol=2e-6, atol=1e-7), ('dropout_apply', seed, case)
        cs = ops.colsum(dG)
        assert np.all(np.abs(cs.cpu().numpy()[:F] - G.astype(np.float64).sum(axis=0)) <= 4e-6 * np.abs(G).sum(axis=0) + 1e-6)
        # feature panels: pack, then unpack, is the identity on the n x F block
        W = int(rng.choice([1, 2, 3, 8]))
        wp = (-(-F // W) + 3) // 4 * 4
        R = n + int(rng.randint(0, 4))
        buf = torch.full((W * R * wp,), 5.0, device=dev)
        ops.pack_panels(dG, R, W, wp, buf)
        v = buf.view(W, R, wp).cpu().numpy()
        full = np.zeros((R, W * wp), np.float32)
        full[:n, :F] = G
        assert np.array_equal(v.transpose(1, 0, 2).reshape(R, W * wp), full), ('pack', seed, case)
        back = ops.DMat.empty(n, F, dev)
        back.t.fill_(3.0)
        ops.unpack_panels(buf, R, W, wp, back)
        assert np.array_equal(back.numpy(), G) and torch.all(back.t[:, F:] == 0), ('unpack', seed, case)


@pytest.mark.parametrize("variant", ["reorder-degree", "reorder-lpa", "reorder-rcm", "reorder-bfs", "reorder-auto", "hip_graph", "bf16x3", "bf16"])
@pytest.mark.parametrize("seed", _seeds(3))
def test_training_step_random_models_variants(dev, seed, variant):
    """The same random models through the optional paths: node reorderings (invisible to the caller: indices, labels and
    outputs stay in ORIGINAL node order), the captured-and-replayed step, the reduced-precision GEMMs."""
    from geographconv_amd.gcnmodel import GraphConv
    from geographconv_amd.nn import layers as L
    from geographconv_amd import synth
    from oracle import gcn_oracle as O
    rng = np.random.RandomState(7500 + seed)
    N = int(rng.choice([40, 150, 333]))
    V = int(rng.choice([11, 40, 97]))
    C = int(rng.choice([3, 7, 33]))
    highway = bool(rng.randint(2))
    depth = int(rng.choice([1, 2, 3]))
    w = int(rng.choice([9, 16, 37]))
    hid = [w] * depth if highway else [int(rng.choice([9, 16, 37])) for _ in range(depth)]
    graphed = variant == 'hip_graph'
    p = 0.0 if graphed else float(rng.choice([0.0, 0.5]))      # (a captured step draws its own Philox mask: compare at p = 0)
    A, X, Y = synth.small_graph(N, 4.0, V, 6, C, seed=seed + 50, empty_rows=1)
    params = O.random_params(V, hid, C, highway, seed=seed + 51, scale=0.5)
    perm = rng.permutation(N)
    tr = np.sort(perm[:N // 2]).astype(np.int32)
    dv = np.sort(perm[N // 2:N // 2 + N // 5]).astype(np.int32)
    mask = (rng.rand(N, hid[0]) < (1 - p)).astype(np.uint8) if p > 0 else np.ones((N, hid[0]), np.uint8)
    kw = {}
    if variant.startswith('reorder-'):
        kw['reorder'] = variant.split('-')[1]
    if variant in ('bf16x3', 'bf16'):
        kw['gemm_precision'] = variant
    clf = GraphConv(V, C, hid, 0.0, p, highway=highway, device=dev, hip_graph=graphed, **kw)
    clf.build_model(A if 'reorder' in kw else None, seed=77)
    L.set_all_param_values(clf.l_out, [q.copy() for q in params])
    if p > 0:
        clf.inject_dropout_mask(mask)
    st = O.AdamState(params)
    cur = [q.copy() for q in params]
    loose = variant == 'bf16'
    n_steps = 4 if graphed else 2                             # (two eager steps, the capture, one replay)
    for step in range(n_steps):
        new, outs, grads = O.f_train(cur, st, X, Y[tr], Y[dv], A, tr, dv, hid, highway, p, mask.astype(np.float32), 0.0)
        o = clf.f_train(X, Y[tr], Y[dv], A, tr, dv)
        what = (variant, seed, dict(N=N, V=V, C=C, hid=hid, highway=highway, p=p), step)
        P = np.asarray(o[4])
        if loose:
            assert np.allclose([float(v) for v in o[:4]], outs[:4], rtol=3e-2, atol=3e-2), (what, o[:4], outs[:4])
            assert np.allclose(P, outs[4], rtol=5e-2, atol=2e-2), what
        else:
            assert np.allclose([float(v) for v in o[:4]], outs[:4], rtol=5e-5, atol=5e-6), (what, o[:4], outs[:4])
            assert np.allclose(P, outs[4], rtol=3e-4, atol=3e-6), what
            for i, (g, r) in enumerate(zip(clf.get_grads(), grads)):
                assert np.allclose(g, r, rtol=1e-3, atol=1e-6 + 4e-5 * np.abs(r).max()), (what, i, np.abs(g - r).max())
        # keep the two trajectories together: continue the restatement from the device's parameters
        cur = [q.copy() for q in L.get_all_param_values(clf.l_out)]
        if not loose:
            for i, (a, b) in enumerate(zip(cur, new)):
                assert np.allclose(a, b, rtol=2e-4, atol=1e-4), (what, 'param', i, np.abs(a - b).max())
        else:
            break                                              # (bf16: one step; Adam would amplify the operand rounding)
    if graphed:
        # a captured step keeps a private memory pool alive: drop the model before the next case builds its own
        import gc
        del clf
        gc.collect()


@pytest.mark.parametrize("seed", _seeds(4))
def test_loss_indexing_and_optimizer_random(dev, seed):
    """Cross-entropy sums / hit counts and the CE gradient on random index sets WITH repeats and gaps (Theano's
    AdvancedSubtensor1 / AdvancedIncSubtensor1 semantics, gcnmodel.py:376-382), row gather / scatter, the Lasagne Adam
    update with L1 + L2 (gcnmodel.py:383-387,407; host and device-resident step counter), the penalty value."""
    from geographconv_amd import ops
    rng = np.random.RandomState(9500 + seed)
    for case in range(6):
        N = int(rng.choice([1, 7, 300, 5000]))
        C = int(rng.choice([2, 5, 129, 256]))
        n_idx = int(rng.choice([0, 1, 5, 400, 7000]))
        L = (rng.randn(N, C) * 2).astype(np.float32)
        e = np.exp(L.astype(np.float64) - L.max(axis=1, keepdims=True))
        P64 = e / e.sum(axis=1, keepdims=True)
        dP = _dmat(ops, dev, P64.astype(np.float32), rng)
        P = dP.numpy().astype(np.float64)
        idx = rng.randint(0, N, size=n_idx).astype(np.int32)                   # repeats on purpose
        y = rng.randint(0, C, size=n_idx).astype(np.int32)
        ti, ty = torch.from_numpy(idx).to(dev), torch.from_numpy(y).to(dev)
        am = torch.from_numpy(P.argmax(axis=1).astype(np.int32)).to(dev)
        for a in (None, am):
            got = ops.ce_metrics(dP, ti, ty, argmax=a).cpu().numpy()
            ref_loss = float(-np.log(P[idx, y]).sum()) if n_idx else 0.0
            ref_hits = float((P.argmax(axis=1)[idx] == y).sum()) if n_idx else 0.0
            assert abs(got[0] - ref_loss) <= 3e-6 * max(1.0, float(np.abs(np.log(P[idx, y])).sum()) if n_idx else 1.0), ('ce loss', seed, case)
            assert got[1] == ref_hits, ('hits', seed, case)
        inv_n = 1.0 / max(1, n_idx)
        ref = np.zeros((N, C))
        np.add.at(ref, idx, (P[idx] - np.eye(C)[y]) * inv_n)
        mag = np.zeros((N, C))
        np.add.at(mag, idx, np.abs(P[idx] - np.eye(C)[y]) * inv_n)
        reps = np.bincount(idx, minlength=N).astype(np.float64)
        db = torch.full((ops.pad4(C),), 3.0, device=dev)
        for with_db in (False, True):
            out = ops.DMat.empty(N, C, dev, ld=ops.gather_ld(C))
            out.t.fill_(7.0)
            got = ops.softmax_ce_bwd(dP, ti, ty, out=out, inv_n=inv_n, **({'db': db} if with_db and C <= 1024 else {}))
            # fp32 accumulation of k repeats of a row: error <= k * eps * (sum of the magnitudes added), as for Theano's
            # AdvancedIncSubtensor1 in fp32
            assert np.all(np.abs(got.numpy() - ref) <= (1.2e-7 * reps[:, None] + 1e-6) * mag + 1e-9), ('ce bwd', seed, case, with_db, N, C, n_idx)
            if with_db:
                assert np.all(np.abs(db.cpu().numpy()[:C] - ref.sum(axis=0)) <= 1e-5 * mag.sum(axis=0) + 2e-6), ('ce db', seed, case)
        if n_idx:
            g = ops.gather_rows(dP, ti)
            assert np.array_equal(g.cpu().numpy(), dP.numpy()[idx])
            uniq = np.unique(idx).astype(np.int32)
            src = _dmat(ops, dev, rng.randn(len(uniq), C).astype(np.float32), rng)
            dst = ops.DMat(N, C, dev)
            ops.scatter_rows(src, torch.from_numpy(uniq).to(dev), dst)
            full = np.zeros((N, C), np.float32)
            full[uniq] = src.numpy()
            assert np.array_equal(dst.numpy(), full)
        # Adam (lasagne.updates.adam) with the L1 + L2 penalty gradient on the regularisable part
        n = int(rng.choice([1, 33, 4096, 100003]))
        p0, g0 = rng.randn(n).astype(np.float32), rng.randn(n).astype(np.float32)
        m0, v0 = (rng.randn(n) * 0.1).astype(np.float32), (rng.rand(n) * 0.1).astype(np.float32)
        mask = (rng.rand(n) < 0.7).astype(np.float32)
        l1, l2 = float(rng.choice([0.0, 1e-3])), float(rng.choice([0.0, 1e-2]))
        t = int(rng.randint(1, 50))
        lr, b1, b2, eps = 2e-3, 0.9, 0.999, 1e-8
        gg = g0.astype(np.float64) + mask * (l1 * np.sign(p0) + 2 * l2 * p0)
        m1 = b1 * m0 + (1 - b1) * gg
        v1 = b2 * v0 + (1 - b2) * gg * gg
        a_t = lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t)
        p1 = p0 - a_t * m1 / (np.sqrt(v1) + eps)
        for ctr in (False, True):
            tp, tg, tm, tv, tk = (torch.from_numpy(a.copy()).to(dev) for a in (p0, g0, m0, v0, mask))
            if ctr:
                state = torch.tensor([t - 1, 0], dtype=torch.int64, device=dev)
                ops.adam_step_ctr(tp, tg, tm, tv, tk, lr, b1, b2, eps, state, l1=l1, l2=l2)
                assert int(state[0].item()) == t
            else:
                ops.adam_step(tp, tg, tm, tv, tk, lr, b1, b2, eps, t, l1=l1, l2=l2)
            assert np.allclose(tp.cpu().numpy(), p1, rtol=2e-5, atol=2e-7), ('adam p', seed, case, ctr)
            assert np.allclose(tm.cpu().numpy(), m1, rtol=1e-5, atol=1e-7), ('adam m', seed, case, ctr)
            assert np.allclose(tv.cpu().numpy(), v1, rtol=1e-5, atol=1e-7), ('adam v', seed, case, ctr)
        pen = ops.reg_penalty(torch.from_numpy(p0).to(dev), torch.from_numpy(mask).to(dev), l1, l2).cpu().numpy()[0]
        ref_pen = float((mask * (l1 * np.abs(p0) + l2 * p0.astype(np.float64) ** 2)).sum())
        assert abs(pen - ref_pen) <= 2e-5 * max(1.0, abs(ref_pen)), ('penalty', seed, case)


@pytest.mark.parametrize("seed", _seeds(3))
def test_medium_sizes_random(dev, seed):
    """The same sweeps at sizes where the launch geometry changes: GEMM rows from a few tiles to thousands (64-row tiles when
    there are fewer tiles than CUs, persistent blocks beyond, split-K slab counts for the transposed products), graph
    products with 10^5 rows (per-XCD row ranges, long-row chunks + ordered combine), the document-blocked X^T . G with
    several batches, X . W0 with the hot rows in LDS."""
    from geographconv_amd import ops, synth
    rng = np.random.RandomState(11000 + seed)
    for case in range(3):
        M = int(rng.choice([4000, 33001, 120000]))
        N, K = int(rng.choice([40, 256, 300])), int(rng.choice([129, 300, 600]))
        A = rng.randn(M, K).astype(np.float32)
        B = (rng.randn(K, N) * 0.1).astype(np.float32)
        G = rng.randn(M, N).astype(np.float32)
        dA, dB, dG = _dmat(ops, dev, A, rng), _dmat(ops, dev, B, rng), _dmat(ops, dev, G, rng)
        a64 = A.astype(np.float64)
        _close(ops.gemm(dA, dB).numpy(), a64 @ B, np.abs(a64) @ np.abs(B), ('gemm nn', seed, case, M, N, K))
        _close(ops.gemm(dG, dB, transB=True).numpy(), G.astype(np.float64) @ B.T, np.abs(G) @ np.abs(B.T), ('gemm nt', seed, case))
        _close(ops.gemm(dA, dG, transA=True).numpy(), a64.T @ G, np.abs(a64.T) @ np.abs(G), ('gemm tn', seed, case, M, N, K), slack=8e-6)
    n = int(rng.choice([60000, 150000]))
    Ah = synth.powerlaw_ahat(n, n * 12, seed=seed)
    F = int(rng.choice([40, 256, 300]))
    Z = rng.randn(n, F).astype(np.float32)
    bias = rng.randn(F).astype(np.float32)
    bt = torch.zeros(ops.pad4(F), device=dev)
    bt[:F] = torch.from_numpy(bias)
    dA = ops.CSR(Ah, dev, long_row_nnz=int(rng.choice([64, 256])), chunk_nnz=int(rng.choice([32, 128])))
    assert dA.n_long_rows > 0
    a64 = Ah.astype(np.float64)
    got = ops.spmm(dA, _dmat(ops, dev, Z, rng), bias=bt, act=1)
    _close(got.numpy(), np.tanh(np.asarray(a64 @ Z.astype(np.float64)) + bias), np.asarray(abs(a64) @ np.abs(Z)) + np.abs(bias),
           ('spmm medium', seed, n, F))
    nd, nw = int(rng.choice([80000, 200000])), int(rng.choice([3000, 9000]))
    X = sps.csr_matrix(synth.bow_x(nd, nw, 40, seed=seed + 3))
    x = ops.SparseOperand.from_scipy(X, dev)
    Fx = int(rng.choice([64, 300]))
    W = (rng.randn(nw, Fx) * 0.1).astype(np.float32)
    Gx = rng.randn(nd, Fx).astype(np.float32)
    x64 = X.astype(np.float64)
    _close(ops.spmm_x(x, _dmat(ops, dev, W, rng)).numpy(), np.asarray(x64 @ W.astype(np.float64)), np.asarray(abs(x64) @ np.abs(W)),
           ('X.W0 medium', seed, nd, nw, Fx))
    _close(ops.spmm_t(x, _dmat(ops, dev, Gx, rng)).numpy(), np.asarray(x64.T @ Gx.astype(np.float64)), np.asarray(abs(x64).T @ np.abs(Gx)),
           ('X^T.G medium', seed, nd, nw, Fx), slack=8e-6)
