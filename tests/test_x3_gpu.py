"""The split-bf16 ("bf16x3") whole-rows and A^T.B kernels (csrc/gemm_x3.hip) through the C ABI against fp64 on the same seeded inputs,
held to the SAME envelope as the exact fp32 MFMA kernels (2e-6 sum|a.b|, tests/test_kernels_gpu.py), and against the exact kernels'
outputs.  Sizes sit just above the 32,768-row threshold from which the whole-rows kernels are taken (run on the MI355X: pytest -m gpu)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def dev():
    from geographconv_amd import ops
    ops.require_gpu()
    return torch.device("cuda:0")


def _rand(shape, seed, scale=1.0):
    return (np.random.RandomState(seed).randn(*shape) * scale).astype(np.float32)


def _wide_range(shape, seed):
    """Values whose magnitudes span 12 decades (gradients next to activations): the split must be exact at every exponent."""
    r = np.random.RandomState(seed)
    return (r.randn(*shape) * 10.0 ** r.uniform(-9, 3, size=shape)).astype(np.float32)


ENV = 2e-6        # the fp32 kernels' envelope: |C - C64| <= ENV * (|A| . |B|) + tiny


def _check(got, ref, mag, what, extra=0.0):
    err = np.abs(got.astype(np.float64) - ref)
    bound = ENV * mag + 1e-30 + extra
    bad = err > bound
    assert not bad.any(), "%s: %d entries beyond the fp32 envelope, worst %.3g x" % (what, bad.sum(), (err / bound).max())


def _takes_x3(M, N, K, transB=False):
    """Is this single product one the split-bf16 whole-rows kernel takes?  (its workspace differs from the staged kernel's)"""
    from geographconv_amd import _ffi
    lib = _ffi.lib()
    return lib.geogcn_gemm_workspace_bytes(0, int(transB), M, N, K, _ffi.GEMM_BF16X3) > 3 * ((K + 31) // 32 * 32) * N * 2


M0 = 33000          # >= 32,768 rows: the whole-rows kernels; not a multiple of 64 (a ragged last tile)


@pytest.mark.parametrize("N,K", [(300, 300), (256, 300), (300, 256), (600, 300), (129, 300), (300, 129), (640, 608), (100, 300),
                                 (900, 900), (930, 900), (900, 256), (1024, 1024), (641, 1000)])
def test_x3_rows_single_products(dev, N, K):
    from geographconv_amd import ops
    A = _wide_range((M0, K), 1)
    B = _rand((K, N), 2, 0.1)
    bias = _rand((N,), 3)
    dA, dB = ops.DMat.from_numpy(A, dev), ops.DMat.from_numpy(B, dev)
    dBt = ops.DMat.from_numpy(np.ascontiguousarray(B.T), dev)
    db = torch.from_numpy(np.pad(bias, (0, ops.pad4(N) - N))).to(dev)
    ref = A.astype(np.float64) @ B.astype(np.float64)
    mag = np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64)
    for kw in (dict(), dict(transB=True)):
        Bop = dBt if kw else dB
        got = ops.gemm(dA, Bop, precision='bf16x3', **kw)
        _check(got.numpy(), ref, mag, 'A.B%s %dx%d' % ('^T' if kw else '', N, K))
        assert np.array_equal(got.numpy(), ops.gemm(dA, Bop, precision='bf16x3', **kw).numpy())          # run to run
        # pad columns [N, ld) written as zeros (geogcn.h convention)
        full = got.t.cpu().numpy().reshape(M0, got.ld)
        assert not full[:, N:ops.pad4(N)].any()
    got = ops.gemm(dA, dB, bias=db, act=ops.ACT_TANH, precision='bf16x3').numpy()
    want = np.tanh(ref + bias)
    assert np.all(np.abs(got - want) <= ENV * mag + 2e-6)
    got = ops.gemm(dA, dB, bias=db, act=ops.ACT_SIGMOID, precision='bf16x3').numpy()
    assert np.all(np.abs(got - 1 / (1 + np.exp(-(ref + bias)))) <= ENV * mag + 2e-6)
    C0 = _rand((M0, N), 4)
    dC = ops.DMat.from_numpy(C0, dev)
    ops.gemm(dA, dB, out=dC, accumulate=True, precision='bf16x3')
    _check(dC.numpy(), ref + C0, mag, 'accumulate', extra=2e-7 * np.abs(ref + C0))


@pytest.mark.parametrize("M", [32768, 32769])
@pytest.mark.parametrize("N,K", [(640, 640), (513, 33), (200, 40), (193, 257)])
def test_x3_rows_boundaries(dev, M, N, K):
    """The threshold row count (a multiple of 64, and one row more: a last tile of ONE row), the widest K and N, a second column pass of
    one tile, a K of one chunk with most of its k-steps past the end."""
    from geographconv_amd import ops
    A = _wide_range((M, K), 11)
    B = _rand((K, N), 12, 0.1)
    dA, dB = ops.DMat.from_numpy(A, dev), ops.DMat.from_numpy(B, dev)
    # the rows the check looks at: the first and last tiles and a sample in between (fp64 of the whole product takes a while at 640 x 640)
    rows = np.unique(np.concatenate([np.arange(0, 130), np.arange(M - 130, M), np.random.RandomState(5).randint(0, M, 500)]))
    ref = A[rows].astype(np.float64) @ B.astype(np.float64)
    mag = np.abs(A[rows]).astype(np.float64) @ np.abs(B).astype(np.float64)
    guard = torch.full((64 * ops.pad4(N),), 7.0, device=dev)          # right behind C in memory most of the time: a ragged tile must not write past row M
    got = ops.gemm(dA, dB, precision='bf16x3')
    _check(got.numpy()[rows], ref, mag, 'A.B %d x %d x %d' % (M, N, K))
    assert bool((guard == 7.0).all())
    full = got.t.cpu().numpy().reshape(M, got.ld)
    assert not full[:, N:ops.pad4(N)].any()
    exact = ops.gemm(dA, dB, precision='f32').numpy()
    assert np.all(np.abs(got.numpy()[rows] - exact[rows]) <= 2 * ENV * mag + 1e-30)
    # (a shape the split-bf16 kernel does not take runs the exact kernels: the same bits)
    assert not np.array_equal(got.numpy(), exact), "the split-bf16 whole-rows kernel did not take %d x %d x %d" % (M, N, K)


def test_x3_non_finite_operands_stay_visible(dev):
    """include/geogcn.h: an Inf / NaN in an operand must come out as a non-finite value in every output it reaches (NaN under bf16x3, where
    the exact kernels give Inf or NaN) and must not touch the other rows."""
    from geographconv_amd import ops
    M, N, K = M0, 300, 300
    A = _rand((M, K), 21)
    B = _rand((K, N), 22, 0.1)
    clean = ops.gemm(ops.DMat.from_numpy(A, dev), ops.DMat.from_numpy(B, dev), precision='bf16x3').numpy()
    A2 = A.copy()
    A2[5, 7] = np.inf
    A2[64 * 100 + 3, 0] = np.nan
    A2[M - 1, K - 1] = -np.inf
    A2[777, 10] = 3.4e38          # finite in fp32, beyond the largest bf16
    got = ops.gemm(ops.DMat.from_numpy(A2, dev), ops.DMat.from_numpy(B, dev), precision='bf16x3').numpy()
    hit = [5, 64 * 100 + 3, M - 1, 777]
    assert not np.isfinite(got[hit]).any()
    rest = np.setdiff1d(np.arange(M), hit)
    assert np.array_equal(got[rest], clean[rest])
    # A^T . B: a non-finite row of A poisons the columns of C it multiplies into -- here all of them in the poisoned row of A^T
    At = _rand((40000, 300), 23)
    Bt = _rand((40000, 300), 24)
    At[123, 17] = np.inf
    gt = ops.gemm(ops.DMat.from_numpy(At, dev), ops.DMat.from_numpy(Bt, dev), transA=True, precision='bf16x3').numpy()
    assert not np.isfinite(gt[17]).any()
    assert np.isfinite(np.delete(gt, 17, axis=0)).all()


def test_x3_rows_shapes_really_take_the_kernel(dev):
    """(the tests above would pass on the staged kernel too: make sure the TwitterUS shapes are routed to the whole-rows kernel)"""
    for (N, K, tb) in [(300, 300, 0), (256, 300, 0), (300, 256, 1), (300, 300, 1)]:
        assert _takes_x3(440000, N, K, bool(tb)), (N, K, tb)
    assert _takes_x3(9475, 300, 300) and not _takes_x3(4095, 300, 300)            # CMU size: taken since round 6 (threshold 4,096 rows)
    # (round 6) the reference's WORLD widths (README.md:177-181: -hid 900 900 900, 930 classes): taken in both orientations
    for (N, K, tb) in [(900, 900, 0), (900, 900, 1), (930, 900, 0), (900, 930, 1), (1024, 1024, 0)]:
        assert _takes_x3(440000, N, K, bool(tb)), (N, K, tb)
    assert not _takes_x3(440000, 1025, 300) and not _takes_x3(440000, 300, 1025)


@pytest.mark.parametrize("N0,N1,K", [(300, 300, 300), (256, 300, 300), (300, 129, 256), (600, 300, 300), (900, 900, 900)])
def test_x3_rows_dual(dev, N0, N1, K):
    """The highway block's forward pair in one launch: (Z, T) = (H . Wh, sigmoid(H . Wt + bt))."""
    from geographconv_amd import ops
    A = _rand((M0, K), 5)
    B0, B1 = _rand((K, N0), 6, 0.1), _rand((K, N1), 7, 0.1)
    b1 = _rand((N1,), 8)
    dA, d0, d1 = (ops.DMat.from_numpy(x, dev) for x in (A, B0, B1))
    db1 = torch.from_numpy(np.pad(b1, (0, ops.pad4(N1) - N1))).to(dev)
    z = ops.DMat.empty(M0, N0, dev, ld=ops.gather_ld(N0))
    Z, T = ops.gemm_dual(dA, d0, d1, out0=z, bias1=db1, act1=ops.ACT_SIGMOID, precision='bf16x3')
    A64 = A.astype(np.float64)
    r0, r1 = A64 @ B0, A64 @ B1
    _check(Z.numpy(), r0, np.abs(A64) @ np.abs(B0), 'dual Z')
    assert np.all(np.abs(T.numpy() - 1 / (1 + np.exp(-(r1 + b1)))) <= ENV * (np.abs(A64) @ np.abs(B1)) + 2e-6)
    # each half is the single product of the same precision, bit for bit (same k order, same epilogue) -- where the pair runs on the
    # split-bf16 kernel at all (a segment it does not take sends the whole launch to the exact kernels)
    if _takes_x3(M0, N0, K) and _takes_x3(M0, N1, K):
        assert np.array_equal(Z.numpy(), ops.gemm(dA, d0, precision='bf16x3').numpy())
    # ... and within the envelope of the exact launch
    Ze, Te = ops.gemm_dual(dA, d0, d1, bias1=db1, act1=ops.ACT_SIGMOID, precision='f32')
    assert np.all(np.abs(Z.numpy() - Ze.numpy()) <= 2 * ENV * (np.abs(A64) @ np.abs(B0)) + 1e-30)
    assert np.abs(T.numpy() - Te.numpy()).max() <= 1e-5


@pytest.mark.parametrize("N,K0,K1", [(300, 300, 300), (300, 256, 300), (256, 300, 300), (129, 300, 129), (600, 600, 600), (900, 900, 900)])
def test_x3_rows_kcat_and_epilogues(dev, N, K0, K1):
    """dH = dZ . Wh^T + dU . Wt^T: plain, accumulating, with the carry gradient, with the dropout + tanh gradient on top."""
    from geographconv_amd import ops
    A0, A1 = _wide_range((M0, K0), 9), _rand((M0, K1), 10)
    W0, W1 = _rand((N, K0), 11, 0.1), _rand((N, K1), 12, 0.1)           # weights as stored: transB
    d = {k: ops.DMat.from_numpy(v, dev) for k, v in dict(A0=A0, A1=A1, W0=W0, W1=W1).items()}
    ref = A0.astype(np.float64) @ W0.T.astype(np.float64) + A1.astype(np.float64) @ W1.T.astype(np.float64)
    mag = np.abs(A0).astype(np.float64) @ np.abs(W0.T) + np.abs(A1).astype(np.float64) @ np.abs(W1.T)
    got = ops.gemm_kcat(d['A0'], d['W0'], d['A1'], d['W1'], transB=True, precision='bf16x3')
    _check(got.numpy(), ref, mag, 'kcat')
    C0 = _rand((M0, N), 13)
    dC = ops.DMat.from_numpy(C0, dev)
    ops.gemm_kcat(d['A0'], d['W0'], d['A1'], d['W1'], out=dC, transB=True, accumulate=True, precision='bf16x3')
    _check(dC.numpy(), ref + C0, mag, 'kcat accumulate', extra=2e-7 * np.abs(ref + C0))
    # carry gradient G * (1 - T) in the epilogue: the stored carry + the accumulating launch, bit for bit
    G, T = _rand((M0, N), 14), np.random.RandomState(15).rand(M0, N).astype(np.float32)
    dG, dT = ops.DMat.from_numpy(G, dev), ops.DMat.from_numpy(T, dev)
    carry = ops.GateCarry(dG, dT)
    fused = ops.gemm_kcat(d['A0'], d['W0'], d['A1'], d['W1'], transB=True, gate_carry=carry, precision='bf16x3')
    two = carry.dense()
    ops.gemm_kcat(d['A0'], d['W0'], d['A1'], d['W1'], out=two, transB=True, accumulate=True, precision='bf16x3')
    assert np.array_equal(fused.numpy(), two.numpy())
    _check(fused.numpy(), ref + G.astype(np.float64) * (1 - T), mag, 'kcat + carry', extra=3e-7 * (np.abs(ref) + np.abs(G)))
    if N % 4 == 0:
        # ... times keep * scale * (1 - Y^2): act_bwd over the gated result, bit for bit
        Y = np.tanh(_rand((M0, N), 16))
        keep = (np.random.RandomState(17).rand(M0, N) < 0.7).astype(np.uint8)
        dY, dk = ops.DMat.from_numpy(Y, dev), torch.from_numpy(keep).to(dev)
        out = ops.DMat.empty(M0, N, dev, ld=ops.gather_ld(N))
        ops.gemm_kcat(d['A0'], d['W0'], d['A1'], d['W1'], out=out, transB=True, gate_carry=carry, tanh_bwd=(dY, dk, 1 / 0.7), precision='bf16x3')
        want = ops.act_bwd(fused, dY, ops.ACT_TANH, keep_mask=dk, scale=1 / 0.7)
        assert np.array_equal(out.numpy(), want.numpy())
    # one product with the carry (the separate launches of the reverse sweep)
    one = ops.gemm(d['A0'], d['W0'], transB=True, precision='bf16x3', gate_carry=carry)
    r1 = A0.astype(np.float64) @ W0.T.astype(np.float64)
    _check(one.numpy(), r1 + G.astype(np.float64) * (1 - T), np.abs(A0).astype(np.float64) @ np.abs(W0.T), 'gated single',
           extra=3e-7 * (np.abs(r1) + np.abs(G)))


@pytest.mark.parametrize("M,N,K", [(300, 300, 70000), (300, 256, 50001), (256, 300, 40000), (129, 300, 33333), (300, 600, 36000),
                                   (900, 900, 40000), (900, 930, 33001), (930, 900, 20000), (512, 1024, 20000)])
def test_x3_tn(dev, M, N, K):
    """dW = H^T . dZ over the node dimension (split-K slabs combined in slab order)."""
    from geographconv_amd import ops
    A, B = _rand((K, M), 18), _wide_range((K, N), 19)
    dA, dB = ops.DMat.from_numpy(A, dev), ops.DMat.from_numpy(B, dev)
    ref = A.T.astype(np.float64) @ B.astype(np.float64)
    mag = np.abs(A.T).astype(np.float64) @ np.abs(B).astype(np.float64)
    got = ops.gemm(dA, dB, transA=True, precision='bf16x3')
    _check(got.numpy(), ref, mag, 'A^T.B')
    assert np.array_equal(got.numpy(), ops.gemm(dA, dB, transA=True, precision='bf16x3').numpy())
    # (x3_tn_kernel really took the shape: the exact kernels give other bits)
    assert not np.array_equal(got.numpy(), ops.gemm(dA, dB, transA=True, precision='f32').numpy())
    C0 = _rand((M, N), 20)
    dC = ops.DMat.from_numpy(C0, dev)
    ops.gemm(dA, dB, out=dC, transA=True, accumulate=True, precision='bf16x3')
    _check(dC.numpy(), ref + C0, mag, 'A^T.B accumulate', extra=2e-7 * np.abs(ref + C0))


def test_x3_tn_dual(dev):
    """(dWh, dWt) = H^T . [dZ | dU] in one launch."""
    from geographconv_amd import ops
    K, M, N0, N1 = 70000, 300, 300, 300
    A, B0, B1 = _rand((K, M), 21), _rand((K, N0), 22, 1e-3), _wide_range((K, N1), 23)
    dA, d0, d1 = (ops.DMat.from_numpy(x, dev) for x in (A, B0, B1))
    g0, g1 = ops.gemm_dual(dA, d0, d1, transA=True, precision='bf16x3')
    A64 = A.T.astype(np.float64)
    _check(g0.numpy(), A64 @ B0, np.abs(A64) @ np.abs(B0), 'dual dW0')
    _check(g1.numpy(), A64 @ B1, np.abs(A64) @ np.abs(B1).astype(np.float64), 'dual dW1')
    # (the single launch cuts the node dimension into other slabs: equal to rounding, not bit for bit)
    one = ops.gemm(dA, d0, transA=True, precision='bf16x3').numpy()
    assert np.all(np.abs(g0.numpy() - one) <= 2 * ENV * (np.abs(A64) @ np.abs(B0)))
    assert np.array_equal(g0.numpy(), ops.gemm_dual(dA, d0, d1, transA=True, precision='bf16x3')[0].numpy())          # run to run


def test_tn_slab_limit_falls_back_to_the_staged_kernel(dev, monkeypatch):
    """A split-K slab that does not fit one buffer descriptor must not run on the kernels that end the slab with the descriptor
    (the guard was taken on the slab before the slab-count cap enlarged it).  The limit is lowered through the library's test seam
    so that small operands reach the fallback; results must not change beyond rounding."""
    from geographconv_amd import _ffi, ops
    K, M, N = 70000, 300, 300
    A, B = _rand((K, M), 24), _rand((K, N), 25)
    dA, dB = ops.DMat.from_numpy(A, dev), ops.DMat.from_numpy(B, dev)
    ref = A.T.astype(np.float64) @ B.astype(np.float64)
    mag = np.abs(A.T).astype(np.float64) @ np.abs(B).astype(np.float64)
    direct = {prec: ops.gemm(dA, dB, transA=True, precision=prec).numpy() for prec in ('f32', 'bf16x3')}
    monkeypatch.setenv('GEOGCN_TN_SLAB_LIMIT', str(64 * 1024))          # the library's test seam (csrc/common.h), read at every call
    for prec in ('f32', 'bf16x3'):
        got = ops.gemm(dA, dB, transA=True, precision=prec).numpy()
        _check(got, ref, mag, 'staged fallback ' + prec)
        assert not np.array_equal(got, direct[prec]), 'the seam did not change the kernel (%s)' % prec
    monkeypatch.delenv('GEOGCN_TN_SLAB_LIMIT')
    for prec in ('f32', 'bf16x3'):
        got = ops.gemm(dA, dB, transA=True, precision=prec).numpy()
        _check(got, ref, mag, 'direct ' + prec)
        assert np.array_equal(got, direct[prec])


@pytest.mark.parametrize("seed", range(6))
def test_x3_random_shapes(dev, seed):
    """Seeded random shapes round the kernels' limits (K and N up to 640, ragged row counts, widths that are no multiple of 4 or 16,
    both weight orientations, with and without bias / accumulate): whatever kernel the library picks, the fp32 envelope holds."""
    from geographconv_amd import ops
    r = np.random.RandomState(1000 + seed)
    M = int(r.randint(32768, 36000))
    K = int(r.choice([r.randint(1, 641), 300, 256, 129, 33]))
    N = int(r.choice([r.randint(1, 641), 300, 256, 304, 17]))
    A, B = _wide_range((M, K), seed), _rand((K, N), seed + 50, 0.2)
    bias = _rand((N,), seed + 99)
    dA = ops.DMat.from_numpy(A, dev)
    transB = bool(r.randint(2))
    dB = ops.DMat.from_numpy(np.ascontiguousarray(B.T) if transB else B, dev)
    db = torch.from_numpy(np.pad(bias, (0, ops.pad4(N) - N))).to(dev)
    ref = A.astype(np.float64) @ B.astype(np.float64)
    mag = np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64)
    _check(ops.gemm(dA, dB, transB=transB, precision='bf16x3').numpy(), ref, mag, 'random A.B %s' % ((M, N, K, transB),))
    got = ops.gemm(dA, dB, transB=transB, bias=db, precision='bf16x3').numpy()
    _check(got, ref + bias, mag, 'random A.B + bias %s' % ((M, N, K, transB),), extra=2e-7 * np.abs(ref + bias))
    # the transposed product over the same operands: (A^T . G)[K x N2], reduction over the M rows
    N2 = int(r.choice([300, 256, 600, r.randint(161, 321)]))
    G = _rand((M, N2), seed + 7, 1e-2)
    dG = ops.DMat.from_numpy(G, dev)
    ref_t = A.T.astype(np.float64) @ G.astype(np.float64)
    _check(ops.gemm(dA, dG, transA=True, precision='bf16x3').numpy(), ref_t, np.abs(A.T).astype(np.float64) @ np.abs(G), 'random A^T.B %s' % ((K, N2, M),))


def test_sparse_inputs_gradient_head_panel_stays_fp32_class_in_the_bf16_configuration(dev, monkeypatch):
    """with the bf16 configuration chosen through the module default (GEOGCN_GEMM_PRECISION=bf16: `precision` arrives as
    None) the dense head panel of X^T . dS0 was multiplied with bf16-rounded operands.  The default is resolved inside spmm_t now: the
    head's product is the split-bf16 one in that configuration, bit for bit what precision='bf16x3' gives, and within the fp32 envelope."""
    import scipy.sparse as sps
    from geographconv_amd import ops
    rng = np.random.RandomState(5)
    n, V, F = 6000, 400, 64
    dense_cols = rng.rand(n, 24) < 0.4                       # 24 columns far denser than the head's threshold
    X = sps.hstack([sps.csr_matrix(dense_cols.astype(np.float32) * rng.rand(n, 24).astype(np.float32)),
                    sps.random(n, V - 24, density=0.01, random_state=rng, dtype=np.float32)]).tocsr().astype(np.float32)
    X.sort_indices()
    op = ops.SparseOperand.from_scipy(X, dev, dense_head=True)
    assert op.head_dense is not None
    G = _rand((n, F), 6)
    dG = ops.DMat.from_numpy(G, dev)
    want = ops.spmm_t(op, dG, precision='bf16x3').numpy()
    monkeypatch.setattr(ops, 'GEMM_PRECISION', 'bf16')
    got_default = ops.spmm_t(op, dG).numpy()
    got_named = ops.spmm_t(op, dG, precision='bf16').numpy()
    assert np.array_equal(got_default, want) and np.array_equal(got_named, want)
    X64 = X.astype(np.float64)
    _check(got_default, np.asarray(X64.T @ G.astype(np.float64)), np.asarray(abs(X64).T @ np.abs(G).astype(np.float64)), 'X^T.G head + tail')


@pytest.mark.parametrize("N,K,W,transB", [(300, 300, 8, False), (300, 300, 3, True), (256, 300, 4, False), (600, 300, 8, False), (320, 256, 5, True)])
def test_x3_rows_panel_output(dev, N, K, W, transB):
    """(round 6) The partitioned path's H . W written as feature panels -- the all-to-all's send layout [W][R][wp] -- by the whole-rows
    split-bf16 kernel itself (until round 5 a bf16x3 panel product ran on the round-1 staged kernel): bit for bit the row-major product
    of the same kernel, rearranged; bias + activation included; pad columns and pad rows untouched."""
    from geographconv_amd import ops
    M, R = M0, M0 + 3
    A = _wide_range((M, K), 31)
    B = _rand((N, K) if transB else (K, N), 32, 0.1)
    bias = _rand((N,), 33)
    dA, dB = ops.DMat.from_numpy(A, dev), ops.DMat.from_numpy(B, dev)
    db = torch.from_numpy(np.pad(bias, (0, ops.pad4(N) - N))).to(dev)
    wp = ops.pad4(-(-N // W))
    plain = None
    for kw in (dict(), dict(bias=db, act=ops.ACT_TANH)):
        rows = ops.gemm(dA, dB, transB=transB, precision='bf16x3', **kw).numpy()
        plain = rows if plain is None else plain
        p = ops.Panels(M, N, R, W, wp, dev)
        p.t.fill_(7.0)
        ops.gemm(dA, dB, out=p, transB=transB, precision='bf16x3', **kw)
        t = p.t.view(W, R, wp).cpu().numpy()
        got = t[:, :M, :].transpose(1, 0, 2).reshape(M, W * wp)
        assert np.array_equal(got[:, :N], rows), (N, K, W, transB, bool(kw))
        assert np.all(t[:, M:, :] == 7.0)                                   # pad rows: untouched
        assert np.all(got[:, ops.pad4(N):] == 7.0)                          # columns beyond roundup4(N): untouched
        assert np.all(got[:, N:ops.pad4(N)] == 0.0)                         # the float4 that holds column N - 1 is completed with zeros
    # the staged split-bf16 kernel (what a panel product ran on before, and still runs on below the row threshold) gives other bits
    small = ops.Panels(1000, N, 1000, W, wp, dev)
    ops.gemm(ops.DMat.from_numpy(A[:1000], dev), dB, out=small, transB=transB, precision='bf16x3')
    gs = small.t.view(W, 1000, wp).cpu().numpy().transpose(1, 0, 2).reshape(1000, W * wp)[:, :N]
    ref = A[:1000].astype(np.float64) @ (B.T if transB else B).astype(np.float64)
    mag = np.abs(A[:1000]).astype(np.float64) @ np.abs(B.T if transB else B).astype(np.float64)
    _check(gs, ref, mag, 'staged panels')
    _check(plain[:1000], ref, mag, 'whole-rows panels')
