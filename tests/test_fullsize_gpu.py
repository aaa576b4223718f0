"""Parity at BASELINE.json's FULL TwitterUS size (N = 440,000, nnz(A_hat) = 10,730,596), where the oracle's
full step takes ~20 s: size-independent properties and sampled exact comparisons instead of a whole-array diff.
  * sampled rows of A.B (incl. every kind of row: short, long/chunked, the 75k-nnz hub) vs a CPU row product;
  * column-sum identity 1^T (A B) = (A^T 1)^T B (a checksum of checksums over all 10.7 M edges);
  * adjointness <u, A v> = <A^T u, v> between the forward and the backward operator;
  * dense contractions on sampled rows / the whole small output;
  * three f_train steps of the 3x300 highway model: first loss = log C for a softmax fed by random weights
    is NOT assumed -- it is compared with the oracle's forward on the same weights at 2,000 sampled rows via the
    row-locality of everything but the SpMM (skipped) -- so here only: finite, and decreasing on repeated steps."""
import numpy as np
import pytest
import scipy.sparse as sps

from geographconv_amd import synth

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

# configs[4] against the bf16-AWARE oracle: absolute + relative bound on every probability (fp32 accumulation-order
# noise + the Z elements it pushes across a bf16 rounding boundary; the first measurement is recorded in DESIGN.md section 3)
BF16_AWARE_P_ATOL, BF16_AWARE_P_RTOL = 1e-5, 1e-3


@pytest.fixture(scope="module")
def twus():
    from geographconv_amd import ops
    ops.require_gpu()
    dev = torch.device("cuda:0")
    A, X, Y, (tr, dv, te), C = synth.make_graph('twus')
    assert synth.check_pinned('twus', 'A', A) and synth.check_pinned('twus', 'X', X)
    return dict(A=A, X=X, Y=Y, tr=tr, dev_idx=dv, te=te, C=C, device=dev, dA=ops.SparseOperand.from_scipy(A, dev))


def _rows_product(A, B, rows):
    """Exact-ish CPU reference for selected output rows, in float64."""
    sub = A[rows].astype(np.float64)
    return np.asarray(sub @ B.astype(np.float64))


@pytest.mark.parametrize("F", [300, 256])
def test_spmm_full_size_sampled_rows_and_checksum(twus, F):
    from geographconv_amd import ops
    A, dev = twus['A'], twus['device']
    rng = np.random.RandomState(F)
    B = rng.randn(A.shape[0], F).astype(np.float32)
    dB = ops.DMat.empty(A.shape[0], F, dev, ld=ops.gather_ld(F))
    dB.t[:, :F].copy_(torch.from_numpy(B))
    bias = rng.randn(F).astype(np.float32)
    tb = torch.from_numpy(bias).to(dev)
    out = ops.spmm(twus['dA'].fwd, dB, bias=tb, act=ops.ACT_TANH).numpy()
    deg = np.diff(A.indptr)
    rows = np.unique(np.r_[rng.randint(0, A.shape[0], 300), np.argsort(-deg)[:40], np.nonzero(deg == deg.min())[0][:20],
                           np.nonzero((deg > 256) & (deg < 300))[0][:20], np.nonzero(deg == 256)[0][:5]])
    ref = np.tanh(_rows_product(A, B, rows) + bias)
    pre_mag = np.asarray(abs(A[rows]).astype(np.float64) @ np.abs(B).astype(np.float64)) + np.abs(bias)
    assert np.all(np.abs(out[rows] - ref) <= 2e-6 * pre_mag + 1e-6), np.abs(out[rows] - ref).max()
    # checksum over every edge: 1^T (A B) == (A^T 1)^T B   (linear part, no epilogue)
    lin = ops.spmm(twus['dA'].fwd, dB).numpy().astype(np.float64)
    colsum_rows = np.asarray(A.astype(np.float64).sum(axis=0)).ravel()       # A^T 1
    ref_cs = colsum_rows @ B.astype(np.float64)
    got_cs = lin.sum(axis=0)
    scale = np.abs(colsum_rows) @ np.abs(B).astype(np.float64)
    assert np.all(np.abs(got_cs - ref_cs) <= 1e-6 * scale), np.abs(got_cs - ref_cs).max()
    # run-to-run determinism at full size
    assert np.array_equal(lin.astype(np.float32), ops.spmm(twus['dA'].fwd, dB).numpy())


def test_spmm_adjointness_forward_backward(twus):
    from geographconv_amd import ops
    A, dev = twus['A'], twus['device']
    rng = np.random.RandomState(5)
    u = rng.randn(A.shape[0], 8).astype(np.float32)
    v = rng.randn(A.shape[0], 8).astype(np.float32)
    Av = ops.spmm(twus['dA'].fwd, ops.DMat.from_numpy(v, dev)).numpy().astype(np.float64)
    Atu = ops.spmm(twus['dA'].bwd, ops.DMat.from_numpy(u, dev)).numpy().astype(np.float64)
    lhs = (u.astype(np.float64) * Av).sum(axis=0)
    rhs = (Atu * v.astype(np.float64)).sum(axis=0)
    assert np.allclose(lhs, rhs, rtol=1e-6, atol=1e-3)
    # 8-column product against scipy on the whole matrix (cheap on the CPU at this width)
    assert np.allclose(Av, np.asarray(A.astype(np.float64) @ v.astype(np.float64)), rtol=1e-5, atol=1e-5)


def test_x_products_full_size(twus):
    from geographconv_amd import ops
    X, dev = twus['X'], twus['device']
    rng = np.random.RandomState(9)
    sx = ops.SparseOperand.from_scipy(X, dev)
    assert sx.head_dense is not None
    W0 = (rng.randn(X.shape[1], 300) * 0.05).astype(np.float32)
    out = ops.spmm(sx.fwd, ops.DMat.from_numpy(W0, dev)).numpy()
    rows = rng.randint(0, X.shape[0], 500)
    assert np.allclose(out[rows], _rows_product(X, W0, rows), rtol=1e-5, atol=2e-6)
    G = rng.randn(X.shape[0], 300).astype(np.float32)
    dW = ops.spmm_t(sx, ops.DMat.from_numpy(G, dev)).numpy()
    words = np.unique(np.r_[rng.randint(0, X.shape[1], 60), sx.head_idx.cpu().numpy()[:20]])
    Xt = sps.csr_matrix(X.T)
    ref = _rows_product(Xt, G, words)
    mag = np.asarray(abs(Xt[words]).astype(np.float64) @ np.abs(G).astype(np.float64))
    assert np.all(np.abs(dW[words] - ref) <= 3e-6 * mag + 1e-5), np.abs(dW[words] - ref).max()


def test_gemm_full_size(twus):
    from geographconv_amd import ops
    dev = twus['device']
    N = twus['A'].shape[0]
    rng = np.random.RandomState(3)
    H = rng.randn(N, 300).astype(np.float32)
    W = (rng.randn(300, 300) * 0.05).astype(np.float32)
    dH, dW = ops.DMat.from_numpy(H, dev), ops.DMat.from_numpy(W, dev)
    rows = np.r_[0, 1, 127, 128, N - 1, rng.randint(0, N, 400)]
    for prec in ('f32', 'bf16x3'):
        Z = ops.gemm(dH, dW, precision=prec).numpy()
        ref = H[rows].astype(np.float64) @ W.astype(np.float64)
        assert np.all(np.abs(Z[rows] - ref) <= 2e-6 * (np.abs(H[rows]) @ np.abs(W)) + 1e-6), prec
    G = rng.randn(N, 300).astype(np.float32)
    dWg = ops.gemm(dH, ops.DMat.from_numpy(G, dev), transA=True).numpy()        # 300 x 300, reduction over N
    ref = H.astype(np.float64).T @ G.astype(np.float64)
    assert np.all(np.abs(dWg - ref) <= 3e-6 * (np.abs(H).T @ np.abs(G)) + 1e-4)


def test_f_train_full_size_is_finite_and_learns(twus):
    from geographconv_amd.gcnmodel import GraphConv
    t = twus
    clf = GraphConv(t['X'].shape[1], t['C'], [300, 300, 300], 0.0, 0.5, highway=True)
    clf.build_model(t['A'], seed=77)
    losses = []
    for _ in range(4):
        o = clf.f_train(t['X'], t['Y'][t['tr']], t['Y'][t['dev_idx']], t['A'], t['tr'], t['dev_idx'])
        losses.append(float(o[0]))
        assert np.isfinite(o[0]) and np.isfinite(o[2]) and 0.0 <= o[1] <= 1.0
    assert abs(losses[0] - np.log(t['C'])) < 0.2          # Glorot init: near-uniform softmax
    assert losses[-1] < losses[0]
    P = np.asarray(o[4])
    assert P.shape == (t['A'].shape[0], t['C']) and np.allclose(P.sum(axis=1), 1.0, atol=1e-4)
    pred, probs = clf.predict(t['X'], t['A'], t['te'][:1000])
    assert pred.shape == (1000,) and np.array_equal(pred, probs.argmax(-1))


_ORACLE_RESULTS = {}


def _oracle_once(key, compute):
    """The oracle's side of a test that runs under both GEMM precisions is the same computation on the same seeded inputs both times
    (15-20 s of CPU at this size): computed for the first leg, kept for the second."""
    if key not in _ORACLE_RESULTS:
        _ORACLE_RESULTS[key] = compute()
    return _ORACLE_RESULTS[key]


@pytest.mark.usefixtures('both_gemm_precisions')
def test_forward_full_size_matches_oracle_argmax_exact(twus):
    """BASELINE configs[2] at its FULL size against the oracle itself (deterministic forward of the 3x300 highway GCN
    over all 440,000 nodes, ~15 s of CPU): probabilities within the stated fp32 tolerance on every row, argmax labels
    bit-exact wherever the oracle's own top-2 margin exceeds that tolerance (rows whose two best classes differ by
    less than the fp32 noise have no well-defined label in either implementation)."""
    from geographconv_amd.gcnmodel import GraphConv
    from geographconv_amd.nn import layers as L
    from oracle import gcn_oracle as O
    t = twus
    hid = [300, 300, 300]
    params = O.random_params(t['X'].shape[1], hid, t['C'], True, seed=7)
    clf = GraphConv(t['X'].shape[1], t['C'], hid, 0.0, 0.5, highway=True)
    clf.build_model(t['A'], seed=77)
    L.set_all_param_values(clf.l_out, params)
    N = t['A'].shape[0]
    pred, probs = clf.predict(t['X'], t['A'], np.arange(N, dtype=np.int32))
    ref = _oracle_once('fwd32', lambda: O.forward(params, t['X'], t['A'], hid, True, dtype=np.float32)['P'])
    # stated fp32 tolerance: 2e-6 absolute + 3e-5 relative.  The relative part is for the hub rows: a row of A_hat with
    # 44,684 stored edges is a 44,684-term fp32 sum, added in chunked order here and strictly sequentially by scipy -- the
    # two fp32 results differ by ~1e-5 relative there (measured 7.3e-6 on a 0.63 probability); 99.9 % of the rows agree to 1e-8
    tol = 2e-6
    err = np.abs(probs - ref)
    worst = np.unravel_index(err.argmax(), err.shape)
    deg = np.diff(t['A'].indptr)
    assert np.all(err.max(1) <= tol + 3e-5 * ref.max(1)), (float(err.max()), worst, int(deg[worst[0]]), float(ref[worst]))
    assert np.percentile(err.max(1), 99.9) <= 1e-7
    top2 = np.partition(ref, -2, axis=1)[:, -2:]
    clear = (top2[:, 1] - top2[:, 0]) > 2 * (tol + 3e-5 * top2[:, 1])
    assert clear.mean() > 0.99
    a32 = ref.argmax(-1)
    assert np.array_equal(pred[clear], a32[clear])
    # the rows INSIDE the tie band are arbitrated by the fp64 oracle (as the CMU test does): every label the HIP path
    # emits is the fp32 oracle's or the fp64 oracle's, and against the fp64 arbiter the HIP path is no worse than the
    # fp32 oracle itself
    ref64 = _oracle_once('fwd64', lambda: O.forward(params, t['X'], t['A'], hid, True, dtype=np.float64)['P'])
    a64 = ref64.argmax(-1)
    either = (pred == a32) | (pred == a64)
    n_hip, n_cpu32 = int((pred != a64).sum()), int((a32 != a64).sum())
    print("TWUS argmax: %d of %d rows inside the tie band; label mismatches vs the fp64 oracle: HIP %d, fp32 oracle %d; "
          "HIP vs fp32 oracle %d" % (int((~clear).sum()), N, n_hip, n_cpu32, int((pred != a32).sum())))
    assert either.all(), int((~either).sum())
    assert n_hip == 0 and n_cpu32 == 0                    # measured: 0 / 0 / 0
    assert np.abs(probs - ref64).max() <= tol + 3e-5


@pytest.mark.usefixtures('both_gemm_precisions')
def test_train_step_full_size_matches_oracle(twus):
    """One complete f_train step (forward, backward, Adam) at the FULL TwitterUS shape against the oracle's step on the
    same (A, X, Y), same parameters, same injected dropout mask: losses, hit counts, every gradient and the updated
    parameters (~20 s of CPU for the oracle)."""
    from geographconv_amd.gcnmodel import GraphConv
    from geographconv_amd.nn import layers as L
    from oracle import gcn_oracle as O
    t = twus
    hid = [300, 300, 300]
    N = t['A'].shape[0]
    params = O.random_params(t['X'].shape[1], hid, t['C'], True, seed=7)
    mask = (np.random.RandomState(3).rand(N, 300) < 0.5).astype(np.uint8)
    clf = GraphConv(t['X'].shape[1], t['C'], hid, 0.0, 0.5, highway=True)
    clf.build_model(t['A'], seed=77)
    L.set_all_param_values(clf.l_out, params)
    clf.inject_dropout_mask(mask)
    ytr, ydv = t['Y'][t['tr']], t['Y'][t['dev_idx']]
    out = clf.f_train(t['X'], ytr, ydv, t['A'], t['tr'], t['dev_idx'])
    new, ref, grads = _oracle_once('step', lambda: O.f_train(params, O.AdamState(params), t['X'], ytr, ydv, t['A'], t['tr'], t['dev_idx'], hid,
                                                             True, 0.5, mask.astype(np.float32)))
    assert abs(out[0] - ref[0]) <= 2e-6 * abs(ref[0]) and abs(out[2] - ref[2]) <= 2e-6 * abs(ref[2])
    assert out[1] == ref[1] and out[3] == ref[3]                                  # hit counts: identical
    for i, (g, r) in enumerate(zip(clf.get_grads(), grads)):
        assert np.abs(g - r).max() <= 2e-4 * np.abs(r).max() + 1e-10, (i, float(np.abs(g - r).max()), float(np.abs(r).max()))
    for i, (q, r) in enumerate(zip(L.get_all_param_values(clf.l_out), new)):
        assert np.abs(q - r).max() <= 2e-3 * 0.05 + 1e-7, i        # Adam's normalised step (see test_e2e_gpu)
        assert np.mean(np.abs(q - r)) <= 2e-6


def test_config5_shape_bf16_six_layers_600_hidden_full_size(twus):
    """BASELINE configs[4]'s shape on one GPU: 6 x 600 highway GCN at the FULL TwitterUS size with bf16 H.W products
    (fp32 accumulate) and the bf16 gathered operand, deterministic forward over all 440,000 nodes.

    The pass / fail comparison is against the bf16-AWARE oracle (`O.forward(..., gemm_operands='bf16')`: operands of every
    H.W rounded to bf16 RNE, fp32 accumulation, Z stored as bf16 -- the same roundings at the same places), so the
    tolerance is an fp32-class one: what is left between the two is the fp32 accumulation order of the MFMA chain and of
    the hub rows' chunked sums, plus the rare Z element that lands on the other side of a bf16 rounding boundary because of
    it.  The statistics against the plain fp32 oracle are printed as a report (they say how far bf16 is from fp32, not
    whether the kernels are right) and bounded loosely."""
    from geographconv_amd.gcnmodel import GraphConv
    from geographconv_amd.nn import layers as L
    from oracle import gcn_oracle as O
    t = twus
    hid = [600] * 6
    params = O.random_params(t['X'].shape[1], hid, t['C'], True, seed=11)
    clf = GraphConv(t['X'].shape[1], t['C'], hid, 0.0, 0.5, highway=True, gemm_precision='bf16')
    clf.build_model(t['A'], seed=77)
    L.set_all_param_values(clf.l_out, params)
    N = t['A'].shape[0]
    idx = np.arange(N, dtype=np.int32)
    pred, probs = clf.predict(t['X'], t['A'], idx)
    assert np.all(np.isfinite(probs)) and np.allclose(probs.sum(1), 1.0, atol=1e-4)
    # -- the gate: against the bf16-aware restatement ----------------------------------------------------------------
    refb = O.forward(params, t['X'], t['A'], hid, True, dtype=np.float32, gemm_operands='bf16')['P']
    errb = np.abs(probs - refb)
    tol = BF16_AWARE_P_ATOL + BF16_AWARE_P_RTOL * refb.max(1)
    top2 = np.partition(refb, -2, axis=1)[:, -2:]
    clear = (top2[:, 1] - top2[:, 0]) > 2 * tol
    n_bad = int((pred[clear] != refb.argmax(-1)[clear]).sum())
    print("config5 shape vs the bf16-aware oracle: max |dP| %.3g, 99.9th pct %.3g, mean %.3g; %d of %d rows inside the tie "
          "band; label mismatches outside it: %d" % (errb.max(), np.percentile(errb.max(1), 99.9), errb.mean(),
                                                     int((~clear).sum()), N, n_bad))
    assert np.all(errb.max(1) <= tol), float(errb.max())
    assert clear.mean() > 0.99 and n_bad == 0
    # -- report: distance of the bf16 configuration from exact fp32 --------------------------------------------------
    ref = O.forward(params, t['X'], t['A'], hid, True, dtype=np.float32)['P']
    err = np.abs(probs - ref)
    kl = (ref * (np.log(ref + 1e-30) - np.log(probs + 1e-30))).sum(1)
    agree = float((pred == ref.argmax(-1)).mean())
    print("config5 shape, bf16 vs fp32 oracle (report): max |dP| %.3g, mean |dP| %.3g, max KL %.3g, label agreement %.4f"
          % (err.max(), err.mean(), kl.max(), agree))
    assert err.max() <= 2e-3 and err.mean() <= 1e-6 and kl.max() <= 2e-5 and agree >= 0.9995      # 2x the measured values
    pred2, probs2 = clf.predict(t['X'], t['A'], idx)
    assert np.array_equal(pred, pred2) and np.array_equal(probs, probs2)


def test_operands_beyond_4gb_use_64bit_offsets():
    """N x F matrices of 4.9 GB (1.2 M rows x 1,024 columns: the reference's WORLD configuration runs 900 hidden units over
    ~1.4 M users, README.md:180): every kernel family must address rows beyond the 2^32-byte mark correctly.  Operands are
    generated on the device; rows sampled from the whole range -- above all from the far end -- are checked on the host."""
    from geographconv_amd import ops, synth
    ops.require_gpu()
    dev = torch.device('cuda:0')
    N, F, K = 1_200_000, 1024, 64
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    Z = ops.DMat.empty(N, F, dev)
    Z.t.normal_(generator=g)
    assert Z.t.numel() * 4 > 2 ** 32
    rows = np.unique(np.r_[0, 1, N // 2, np.arange(N - 40, N), np.random.RandomState(0).randint(0, N, 60)])
    # graph product: a sparse matrix whose far rows reference far rows
    A = synth.powerlaw_ahat(N, 3 * N, seed=2)
    dA = ops.CSR(A, dev)
    S = ops.spmm(dA, Z)
    Zh = {}
    need = np.unique(np.concatenate([A.indices[A.indptr[r]:A.indptr[r + 1]] for r in rows]))
    Zsub = Z.t[torch.from_numpy(need).to(dev)].cpu().numpy().astype(np.float64)
    pos = {int(c): i for i, c in enumerate(need)}
    Sg = S.t[torch.from_numpy(rows).to(dev)].cpu().numpy()
    for i, r in enumerate(rows):
        cols, vals = A.indices[A.indptr[r]:A.indptr[r + 1]], A.data[A.indptr[r]:A.indptr[r + 1]].astype(np.float64)
        ref = (vals[:, None] * Zsub[[pos[int(c)] for c in cols]]).sum(axis=0)
        mag = (np.abs(vals)[:, None] * np.abs(Zsub[[pos[int(c)] for c in cols]])).sum(axis=0)
        assert np.all(np.abs(Sg[i] - ref) <= 4e-6 * mag + 1e-6), ('spmm', int(r))
    del S
    # GEMM: (N x K) . (K x F) -> N x F beyond 4 GB, and the transposed product reducing over all N rows
    H = ops.DMat.empty(N, K, dev)
    H.t.normal_(generator=g)
    W = ops.DMat.empty(K, F, dev)
    W.t.normal_(generator=g)
    C = ops.gemm(H, W, act=ops.ACT_TANH)
    Hr = H.t[torch.from_numpy(rows).to(dev)].cpu().numpy().astype(np.float64)
    Wn = W.numpy().astype(np.float64)
    Cg = C.t[torch.from_numpy(rows).to(dev)].cpu().numpy()
    assert np.all(np.abs(Cg - np.tanh(Hr @ Wn)) <= 4e-6 * (np.abs(Hr) @ np.abs(Wn)) + 1e-6)
    dW = ops.gemm(H, C, transA=True)                                   # K x F, reduction over 1.2 M rows of a 4.9 GB operand
    ref = (H.t.double().T @ C.t.double()).cpu().numpy()
    mag = (H.t.double().abs().T @ C.t.double().abs()).cpu().numpy()
    assert np.all(np.abs(dW.numpy() - ref) <= 8e-6 * mag + 1e-5)
    # elementwise family: highway mix and its backward with the column sums on the same 4.9 GB operands
    T = ops.DMat.empty(N, F, dev)
    T.t.uniform_(generator=g)
    out = ops.highway_fwd(T, C, Z)
    sel = torch.from_numpy(rows).to(dev)
    t_, c_, z_ = (m.t[sel].double().cpu().numpy() for m in (T, C, Z))
    assert np.allclose(out.t[sel].cpu().numpy(), t_ * c_ + (1 - t_) * z_, rtol=2e-6, atol=1e-6)
    dbS, dbU = torch.zeros(F, device=dev), torch.zeros(F, device=dev)
    dS, dU, dC = ops.highway_bwd(out, T, C, Z, dbS=dbS, dbU=dbU)
    g_ = out.t[sel].double().cpu().numpy()
    assert np.allclose(dS.t[sel][:, :F].cpu().numpy(), g_ * t_ * (1 - c_ * c_), rtol=3e-6, atol=1e-6)
    assert np.allclose(dC.t[sel].cpu().numpy(), g_ * (1 - t_), rtol=3e-6, atol=1e-6)
    ref_db = (out.t.double() * T.t.double() * (1 - C.t.double() ** 2)).sum(dim=0).cpu().numpy()
    assert np.allclose(dbS.cpu().numpy(), ref_db, rtol=2e-4, atol=2e-2)
