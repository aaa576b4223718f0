"""Kernel-level parity: every C-ABI entry point of libgeogcn.so vs the NumPy oracle on the same
seeded inputs (run on the MI355X: pytest -m gpu).  fp32 tolerances are stated per test; integer
outputs (argmax, masks, hit counts) are compared exactly."""
import numpy as np
import pytest
import scipy.sparse as sps

from geographconv_amd import synth, tuning
from oracle import gcn_oracle as O

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(autouse=True)
def _exact_products_by_default(monkeypatch):
    """This file holds the EXACT fp32 kernels to the oracle and to each other bit for bit: products whose precision a test does not
    name run 'f32' here (the library's default is 'bf16x3'; the split-bf16 kernels have tests/test_x3_gpu.py)."""
    from geographconv_amd import ops
    monkeypatch.setattr(ops, 'GEMM_PRECISION', 'f32')


@pytest.fixture(scope="module")
def dev():
    from geographconv_amd import ops
    ops.require_gpu()
    return torch.device("cuda:0")


def _rand(shape, seed, scale=1.0):
    return (np.random.RandomState(seed).randn(*shape) * scale).astype(np.float32)


def _skewed_csr(n_rows, n_cols, seed, hub_nnz=3000, empty_every=7):
    """Random CSR with empty rows, 1-nnz rows and one hub row far above the split threshold."""
    rng = np.random.RandomState(seed)
    rows, cols = [], []
    for r in range(n_rows):
        if r % empty_every == 3:
            k = 0
        elif r == n_rows // 2:
            k = min(hub_nnz, n_cols)
        elif r % 11 == 5:
            k = min(300, n_cols)               # just above the default long-row threshold (256)
        else:
            k = min(n_cols, 1 + rng.poisson(12))
        c = rng.choice(n_cols, size=k, replace=False)
        rows += [r] * k
        cols += list(c)
    vals = rng.randn(len(rows)).astype(np.float32)
    m = sps.csr_matrix((vals, (rows, cols)), shape=(n_rows, n_cols), dtype=np.float32)
    m.sort_indices()
    return m


@pytest.mark.parametrize("F", [1, 5, 64, 129, 256, 300, 600, 900])
@pytest.mark.parametrize("act,use_bias", [(0, False), (1, True)])
def test_spmm_matches_oracle(dev, F, act, use_bias):
    from geographconv_amd import ops
    A = _skewed_csr(700, 900, seed=F)
    B = _rand((900, F), 1)
    bias = _rand((F,), 2) if use_bias else None
    ref = O.spmm(A, B)
    if bias is not None:
        ref = ref + bias
    if act == 1:
        ref = np.tanh(ref)
    dA = ops.CSR(A, dev)
    assert dA.n_long_rows >= 1 and dA.n_chunks > dA.n_long_rows
    dB = ops.DMat.from_numpy(B, dev)
    db = torch.from_numpy(np.pad(bias, (0, ops.pad4(F) - F))).to(dev) if use_bias else None
    out = ops.spmm(dA, dB, bias=db, act=act)
    got = out.numpy()
    # row sums of up to 3000 fp32 products: tolerance scales with sum |a||b|
    mag = np.asarray(abs(A) @ np.abs(B)) + (np.abs(bias) if use_bias else 0)
    assert np.all(np.abs(got - ref) <= 2e-6 * mag + 1e-6), np.abs(got - ref).max()
    # pad columns stay zero
    assert torch.all(out.t[:, F:] == 0)
    # empty rows give act(bias)
    empty = np.diff(A.indptr) == 0
    assert empty.any()


@pytest.mark.parametrize("F", [5, 64, 129, 300, 600])
def test_spmm_bf16_gathered_operand(dev, F):
    """bf16 configuration (BASELINE config 5): the gathered operand is rounded to bfloat16 (RNE, bit-identical
    to torch's cast), products and the sequential accumulation stay fp32 -- so the result equals the fp32 SpMM
    applied to the rounded operand."""
    from geographconv_amd import ops
    A = _skewed_csr(700, 900, seed=F)
    B = _rand((900, F), 1)
    B[0, 0] = np.float32(1.00390625)          # exactly between two bf16 values: ties to even
    B[1, 0] = np.float32(1.01171875)
    bias = _rand((F,), 2)
    dA = ops.CSR(A, dev)
    dB = ops.DMat.from_numpy(B, dev)
    hB = ops.cast_bf16(dB)
    assert hB.ld % 64 == 0 and hB.t.dtype == torch.bfloat16
    want = torch.from_numpy(B).to(torch.bfloat16)
    assert torch.equal(hB.t[:, :F].cpu(), want)
    assert torch.all(hB.t[:, F:].float() == 0)
    Br = want.float().numpy()
    db = torch.from_numpy(np.pad(bias, (0, ops.pad4(F) - F))).to(dev)
    out = ops.spmm(dA, hB, bias=db, act=1)
    ref = np.tanh(O.spmm(A, Br) + bias)
    mag = np.asarray(abs(A) @ np.abs(Br)) + np.abs(bias)
    assert np.all(np.abs(out.numpy() - ref) <= 2e-6 * mag + 1e-6)
    assert torch.all(out.t[:, F:] == 0)
    # and it is the SAME arithmetic as the fp32 kernel on the rounded operand: bitwise equal
    same = ops.spmm(dA, ops.DMat.from_numpy(Br, dev), bias=db, act=1)
    assert torch.equal(same.t, out.t)
    # a pitch that is not a multiple of 4 elements is refused (no scalar fallback for the bf16 operand)
    from geographconv_amd import _ffi
    ws = dA._ws.get(_ffi.lib().geogcn_spmm_workspace_bytes(dA._plan, F))
    rc = _ffi.lib().geogcn_spmm_csr_bf16b(dA._plan, 700, 900, A.nnz, ops._p(dA.rowptr), ops._p(dA.colidx), ops._p(dA.val),
                                         ops._p(hB.t), hB.ld - 2, ops._p(out.t), out.ld, min(F, hB.ld - 2), None, 0,
                                         ops._p(ws), ws.numel(), ops._stream())
    assert rc == -3


@pytest.mark.parametrize("F,bf", [(300, False), (129, False), (64, False), (300, True), (600, True)])
def test_spmm_highway_epilogue(dev, F, bf):
    """geogcn_spmm_csr_highway_f32 = geogcn_spmm_csr_f32(tanh) followed by geogcn_highway_fwd_f32, element for
    element (long rows go through the reduce kernel's copy of the epilogue)."""
    from geographconv_amd import ops
    A = _skewed_csr(700, 700, seed=F)
    dA = ops.CSR(A, dev)
    assert dA.n_long_rows >= 1
    Z = ops.DMat.from_numpy(_rand((700, F), 1), dev)
    if bf:
        Z = ops.cast_bf16(Z)
    bias = torch.from_numpy(np.pad(_rand((F,), 2), (0, ops.pad4(F) - F))).to(dev)
    T = ops.DMat.from_numpy(O.sigmoid(_rand((700, F), 3)).astype(np.float32), dev)
    H = ops.DMat.from_numpy(_rand((700, F), 4), dev)
    Hc, Hout = ops.spmm_highway(dA, Z, bias, T, H)
    Hc_ref = ops.spmm(dA, Z, bias=bias, act=ops.ACT_TANH, F=F)
    Hout_ref = ops.highway_fwd(T, Hc_ref, H)
    assert torch.equal(Hc.t, Hc_ref.t)
    assert torch.allclose(Hout.t, Hout_ref.t, rtol=1e-6, atol=1e-6)      # (fma contraction may differ by an ulp)
    assert torch.all(Hout.t[:, F:] == 0) and torch.all(Hc.t[:, F:] == 0)


@pytest.mark.parametrize("F", [300, 129, 7])
def test_spmm_bias_read_as_quads_or_scalars_gives_the_same_bits(dev, F):
    """(round 6) The epilogues read their bias as one 16-byte load per float4 of the row where the quad lies inside the bias and the
    vector is 16-byte aligned (csrc/common.h load_bias4), as scalars otherwise: a bias vector at an unaligned address (a view one
    float into a buffer, exactly F floats long) gives the bits of the aligned, padded one -- plain product, highway epilogue,
    softmax epilogue, the bf16 operand."""
    from geographconv_amd import ops
    A = _skewed_csr(3000, 2000, 5, hub_nnz=900)
    dA = ops.CSR(A, dev)
    B = ops.DMat.from_numpy(_rand((2000, F), 6), dev)
    bias = _rand((F,), 7)
    aligned = torch.zeros(ops.pad4(F) + 4, device=dev)
    aligned[:F] = torch.from_numpy(bias).to(dev)
    shifted = torch.full((F + 9,), float('nan'), device=dev)
    shifted[1:1 + F] = torch.from_numpy(bias).to(dev)
    un = shifted[1:1 + F]                                  # 4 bytes past a 16-byte boundary; NaN right behind its last element
    assert un.data_ptr() % 16 == 4 and aligned.data_ptr() % 16 == 0
    for Bop in (B, ops.cast_bf16(B)):
        a = ops.spmm(dA, Bop, bias=aligned, act=ops.ACT_TANH)
        b = ops.spmm(dA, Bop, bias=un, act=ops.ACT_TANH)
        assert torch.equal(a.t, b.t) and torch.isfinite(b.t).all()
    T = ops.DMat.from_numpy(O.sigmoid(_rand((3000, F), 8)).astype(np.float32), dev)
    H = ops.DMat.from_numpy(_rand((3000, F), 9), dev)
    h1 = ops.spmm_highway(dA, B, aligned, T, H)
    h2 = ops.spmm_highway(dA, B, un, T, H)
    assert torch.equal(h1[0].t, h2[0].t) and torch.equal(h1[1].t, h2[1].t)
    if ops.spmm_softmax_ok(B, F):
        assert torch.equal(ops.spmm_softmax(dA, B, aligned).t, ops.spmm_softmax(dA, B, un).t)


def test_spmm_short_rows_bitwise_reproducible_and_sequential(dev):
    """Rows below the split threshold accumulate in stored order with fmaf: two runs are bitwise
    equal and equal to a sequential fp32 fma chain (checked via float64 emulation bound)."""
    from geographconv_amd import ops
    A, X, Y = synth.small_graph(3000, 8.0, 400, 20, 7, seed=4)
    B = _rand((3000, 300), 3)
    dA = ops.CSR(A, dev)
    dB = ops.DMat.from_numpy(B, dev)
    o1 = ops.spmm(dA, dB).numpy()
    o2 = ops.spmm(dA, dB).numpy()
    assert np.array_equal(o1, o2)
    ref = O.spmm(A, B)
    assert np.allclose(o1, ref, rtol=1e-5, atol=1e-6)


def test_spmm_long_row_chunks_on_the_owning_xcd_or_dealt_round(dev):
    """geogcn_spmm_plan_create(chunks_with_owner): where the long rows' chunks run is a scheduling decision -- both
    placements give bitwise the same product (plain, with the highway epilogue, bf16 operand, narrow 8-lane variant), for
    hubs at the front of the numbering and for hubs spread over it; ops.CSR picks by measuring the long rows' locality."""
    from geographconv_amd import ops
    A = synth.powerlaw_ahat(30000, 400000)                    # hubs = lowest row ids
    perm = np.random.RandomState(0).permutation(30000)
    for M in (A, sps.csr_matrix(A[perm][:, perm])):
        own, deal = ops.CSR(M, dev, local=True), ops.CSR(M, dev, local=False)
        assert own.n_chunks == deal.n_chunks > 0 and own.chunks_with_owner and not deal.chunks_with_owner
        for F in (300, 24):
            B = ops.DMat.from_numpy(_rand((30000, F), F), dev)
            b = torch.from_numpy(np.pad(_rand((F,), 2), (0, ops.pad4(F) - F))).to(dev)
            assert torch.equal(ops.spmm(own, B, bias=b, act=ops.ACT_TANH).t, ops.spmm(deal, B, bias=b, act=ops.ACT_TANH).t)
        B = ops.DMat.from_numpy(_rand((30000, 300), 5), dev)
        assert torch.equal(ops.spmm(own, ops.cast_bf16(B)).t, ops.spmm(deal, ops.cast_bf16(B)).t)
        T, H = ops.DMat.from_numpy(np.abs(_rand((30000, 300), 6)) % 1.0, dev), ops.DMat.from_numpy(_rand((30000, 300), 7), dev)
        b = torch.from_numpy(_rand((300,), 2)).to(dev)
        r1, r2 = ops.spmm_highway(own, B, b, T, H), ops.spmm_highway(deal, B, b, T, H)
        assert torch.equal(r1[0].t, r2[0].t) and torch.equal(r1[1].t, r2[1].t)
    assert ops.CSR(A, dev).chunks_with_owner is False        # (a power-law graph without community structure)
    band = sps.diags([np.ones(30000 - abs(k)) for k in range(-200, 201)], list(range(-200, 201)), format='csr', dtype=np.float32)
    assert ops.CSR(band, dev, long_row_nnz=256).chunks_with_owner is True


@pytest.mark.parametrize("C", [256, 129, 40, 300, 512])
def test_spmm_with_the_softmax_in_its_epilogue(dev, C):
    """geogcn_spmm_csr_softmax_f32 (the output layer: softmax(A_hat . Z + b) without ever writing the logits) against
    geogcn_spmm_csr_f32 + geogcn_softmax_rows_f32: the same first index of every row's maximum, probabilities to rounding (the row
    sum is taken over another layout), rows summing to one, pad columns zero -- short rows (epilogue of the row kernel) and
    long rows (combined, then softmaxed in place) alike; two runs bitwise equal."""
    from geographconv_amd import ops
    A = synth.powerlaw_ahat(20000, 260000)
    dA = ops.CSR(A, dev)
    assert dA.n_chunks > 0                                   # there are long rows
    Z = ops.DMat.from_numpy(_rand((20000, C), 3, 4.0), dev)
    b = torch.from_numpy(np.pad(_rand((C,), 2), (0, ops.pad4(C) - C))).to(dev)
    am_ref = torch.empty(20000, dtype=torch.int32, device=dev)
    ref = ops.softmax_rows(ops.spmm(dA, Z, bias=b), argmax=am_ref)
    am = torch.full((20000,), -1, dtype=torch.int32, device=dev)
    got = ops.spmm_softmax(dA, Z, bias=b, argmax=am)
    assert torch.equal(am, am_ref)
    assert torch.allclose(got.t[:, :C], ref.t[:, :C], rtol=2e-6, atol=1e-12)
    assert torch.all(got.t[:, C:] == 0) and torch.all((got.t[:, :C].sum(1) - 1).abs() < 1e-5)
    again = ops.spmm_softmax(dA, Z, bias=b)
    assert torch.equal(again.t, got.t)
    assert np.abs(got.numpy() - O.softmax_rows(O.spmm(A, Z.numpy()) + b.cpu().numpy()[:C])).max() < 2e-6


def test_spmm_timer_rides_on_the_plan_handle(dev):
    """The profiling timer is a caller-held handle attached to ONE plan (no library-global state): products on another
    plan, of another width, or fused highway launches are not sampled; detaching stops the sampling."""
    from geographconv_amd import ops
    A = synth.powerlaw_ahat(5000, 60000)
    a1, a2 = ops.CSR(A, dev), ops.CSR(A, dev)
    B = ops.DMat.from_numpy(_rand((5000, 64), 3), dev)
    B2 = ops.DMat.from_numpy(_rand((5000, 32), 4), dev)
    t = ops.SpmmTimer(capacity=8)
    t.attach(a1, only_F=64)
    ops.spmm(a1, B)
    ops.spmm(a2, B)            # other plan
    ops.spmm(a1, B2)           # other width
    ops.spmm(a1, B)
    ms = t.read_ms()
    assert len(ms) == 2 and all(m > 0 for m in ms)
    t.detach()
    ops.spmm(a1, B)
    assert len(t.read_ms()) == 2
    t.attach(a2)               # any width; re-attaching resets the pool
    ops.spmm(a2, B2)
    assert len(t.read_ms()) == 1
    del t                      # detaches before the pool is destroyed
    ops.spmm(a2, B2)
    torch.cuda.synchronize()


def test_spmm_scalar_fallback_for_odd_pitch(dev):
    from geographconv_amd import _ffi, ops
    import ctypes as C
    A = _skewed_csr(200, 150, seed=9, hub_nnz=140)
    B = _rand((150, 7), 5)
    dA = ops.CSR(A, dev)
    tB = torch.from_numpy(B).to(dev)                 # pitch 7: not float4-addressable
    tC = torch.zeros((200, 7), dtype=torch.float32, device=dev)
    lib = _ffi.lib()
    rc = lib.geogcn_spmm_csr_f32(None, 200, 150, A.nnz, ops._p(dA.rowptr), ops._p(dA.colidx), ops._p(dA.val),
                                 ops._p(tB), 7, ops._p(tC), 7, 7, None, 0, None, 0, ops._stream())
    assert rc == 0
    assert np.allclose(tC.cpu().numpy(), O.spmm(A, B), rtol=1e-5, atol=1e-5)


def test_spmm_argument_errors(dev):
    from geographconv_amd import _ffi, ops
    lib = _ffi.lib()
    A = _skewed_csr(50, 50, seed=1, hub_nnz=40)
    dA = ops.CSR(A, dev)
    B = ops.DMat.from_numpy(_rand((50, 8), 1), dev)
    out = ops.DMat(50, 8, dev)
    rc = lib.geogcn_spmm_csr_f32(dA._plan, 50, 50, A.nnz, None, ops._p(dA.colidx), ops._p(dA.val), ops._p(B.t), 8,
                                 ops._p(out.t), 8, 8, None, 0, None, 0, ops._stream())
    assert rc == -1 and b'null' in lib.geogcn_last_error()
    rc = lib.geogcn_spmm_csr_f32(dA._plan, 50, 50, A.nnz, ops._p(dA.rowptr), ops._p(dA.colidx), ops._p(dA.val),
                                 ops._p(B.t), 4, ops._p(out.t), 8, 8, None, 0, None, 0, ops._stream())
    assert rc == -2
    with pytest.raises(ValueError):
        ops.spmm(dA, ops.DMat(49, 8, dev))


@pytest.mark.parametrize("M,N,K", [(1000, 300, 300), (777, 129, 300), (513, 600, 300), (64, 16, 4), (130, 132, 129),
                                   (2500, 256, 300)])
def test_gemm_nn_nt_match_oracle(dev, M, N, K):
    from geographconv_amd import ops
    A = _rand((M, K), 1)
    Bnn = _rand((K, N), 2)
    bias = _rand((N,), 3)
    dA = ops.DMat.from_numpy(A, dev)
    dB = ops.DMat.from_numpy(Bnn, dev)
    db = torch.from_numpy(np.pad(bias, (0, ops.pad4(N) - N))).to(dev)
    ref = A.astype(np.float64) @ Bnn.astype(np.float64)
    tol = 2e-6 * (np.abs(A) @ np.abs(Bnn)) + 1e-6       # fp32 fma-chain error envelope
    got = ops.gemm(dA, dB).numpy()
    assert np.all(np.abs(got - ref) <= tol)
    got = ops.gemm(dA, dB, bias=db, act=ops.ACT_SIGMOID).numpy()
    sref = 1.0 / (1.0 + np.exp(-(ref + bias)))
    # d sigmoid = s(1-s) * d pre-activation, plus 2 ulp of the result
    assert np.all(np.abs(got - sref) <= sref * (1 - sref) * tol + 3e-7 * sref + 1e-30)
    # NT: C = A . B^T with B given as N x K
    dBt = ops.DMat.from_numpy(np.ascontiguousarray(Bnn.T), dev)
    got = ops.gemm(dA, dBt, transB=True).numpy()
    assert np.all(np.abs(got - ref) <= tol)
    # accumulate
    C0 = _rand((M, N), 4)
    dC = ops.DMat.from_numpy(C0, dev)
    ops.gemm(dA, dB, out=dC, accumulate=True)
    assert np.all(np.abs(dC.numpy() - (ref + C0)) <= tol + 1e-6)


@pytest.mark.parametrize("M,N,K", [(1000, 300, 300), (777, 129, 300), (2500, 256, 300), (513, 600, 300), (130, 300, 129)])
def test_gemm_bf16_modes(dev, M, N, K):
    """bf16x3: three-term split, fp32-class accuracy (same envelope as the exact fp32 MFMA path);
    bf16: single term, the accuracy of BASELINE config 5 (relative 2^-8 per operand)."""
    from geographconv_amd import ops
    A = _rand((M, K), 1)
    Bnn = _rand((K, N), 2)
    bias = _rand((N,), 3)
    dA, dB = ops.DMat.from_numpy(A, dev), ops.DMat.from_numpy(Bnn, dev)
    dBt = ops.DMat.from_numpy(np.ascontiguousarray(Bnn.T), dev)
    db = torch.from_numpy(np.pad(bias, (0, ops.pad4(N) - N))).to(dev)
    ref = A.astype(np.float64) @ Bnn.astype(np.float64)
    mag = np.abs(A) @ np.abs(Bnn)
    for kw in (dict(), dict(transB=True)):
        Bop = dBt if kw else dB
        got = ops.gemm(dA, Bop, precision='bf16x3', **kw).numpy()
        assert np.all(np.abs(got - ref) <= 2e-6 * mag + 1e-6), np.abs(got - ref).max()
        got1 = ops.gemm(dA, Bop, precision='bf16', **kw).numpy()
        assert np.all(np.abs(got1 - ref) <= 2 ** -7 * mag + 1e-6)
        assert np.abs(got1 - ref).max() > 1e-4                      # it really is the coarse mode
    got = ops.gemm(dA, dB, bias=db, act=ops.ACT_TANH, precision='bf16x3').numpy()
    assert np.all(np.abs(got - np.tanh(ref + bias)) <= 2e-6 * mag + 1e-6)
    C0 = _rand((M, N), 4)
    dC = ops.DMat.from_numpy(C0, dev)
    ops.gemm(dA, dB, out=dC, accumulate=True, precision='bf16x3')
    assert np.all(np.abs(dC.numpy() - (ref + C0)) <= 2e-6 * mag + 2e-6)
    assert np.array_equal(ops.gemm(dA, dB, precision='bf16x3').numpy(), ops.gemm(dA, dB, precision='bf16x3').numpy())
    # bf16 result (geogcn_gemm_f32_bf16c): exactly the round-to-nearest-even of the fp32 result of the same mode;
    # whole 8-column pieces written, pads as zeros
    for kw in (dict(), dict(transB=True)):
        Bop = dBt if kw else dB
        f32 = ops.gemm(dA, Bop, bias=db, act=ops.ACT_TANH, precision='bf16', **kw)
        h = ops.HMat(M, N, dev)
        h.t.fill_(7.0)
        ops.gemm(dA, Bop, out=h, bias=db, act=ops.ACT_TANH, precision='bf16', **kw)
        assert torch.equal(h.t[:, :N], f32.t[:, :N].to(torch.bfloat16))
        n8 = (N + 7) // 8 * 8
        assert torch.all(h.t[:, N:n8].float() == 0)
    with pytest.raises(ValueError):
        ops.gemm(dA, dB, out=ops.HMat(M, N, dev), precision='f32')


@pytest.mark.parametrize("M,N,K", [(1000, 600, 600), (130, 640, 600), (70, 256, 600), (333, 300, 256), (65, 7, 300),
                                   (4099, 321, 300), (64, 320, 590), (1, 600, 600)])
def test_gemm_bf16_whole_rows_kernel(dev, M, N, K):
    """The skinny shapes of the bf16 configuration (K padded to 256 / 320 / 608, N <= 640) run on gemm_bf16_rows_kernel: 64
    whole rows of A per block, weights in fragment order.  Checked against the SAME arithmetic in float64 -- operands rounded
    to bf16 (RNE), exact products -- within the fp32 accumulation envelope, through every epilogue (bias + activation,
    accumulate, bf16 result), both weight layouts, ragged row counts, and for run-to-run equality."""
    from geographconv_amd import ops

    def bf16(x):
        return torch.from_numpy(np.ascontiguousarray(x)).to(torch.bfloat16).double().numpy()

    A = _rand((M, K), 11)
    W = _rand((K, N), 12) * 0.3
    bias = _rand((N,), 13)
    dA, dW = ops.DMat.from_numpy(A, dev), ops.DMat.from_numpy(W, dev)
    dWt = ops.DMat.from_numpy(np.ascontiguousarray(W.T), dev)
    db = torch.from_numpy(np.pad(bias, (0, ops.pad4(N) - N))).to(dev)
    ref = bf16(A) @ bf16(W)
    env = 4e-7 * K ** 0.5 * (np.abs(bf16(A)) @ np.abs(bf16(W))) + 1e-6
    for kw in (dict(), dict(transB=True)):
        Bop = dWt if kw else dW
        got = ops.gemm(dA, Bop, precision='bf16', **kw)
        assert np.all(np.abs(got.numpy() - ref) <= env), np.abs(got.numpy() - ref).max()
        assert torch.all(got.t[:, N:] == 0)                                     # pad columns stay zero
        assert torch.equal(got.t, ops.gemm(dA, Bop, precision='bf16', **kw).t)
        got = ops.gemm(dA, Bop, bias=db, act=ops.ACT_SIGMOID, precision='bf16', **kw).numpy()
        want = 1.0 / (1.0 + np.exp(-(ref + bias)))
        assert np.all(np.abs(got - want) <= env + 1e-6)
        C0 = _rand((M, N), 14)
        dC = ops.DMat.from_numpy(C0, dev)
        ops.gemm(dA, Bop, out=dC, accumulate=True, precision='bf16', **kw)
        assert np.all(np.abs(dC.numpy() - (ref + C0)) <= env + 1e-6)
        f32 = ops.gemm(dA, Bop, bias=db, act=ops.ACT_TANH, precision='bf16', **kw)
        h = ops.HMat(M, N, dev)
        h.t.fill_(7.0)
        ops.gemm(dA, Bop, out=h, bias=db, act=ops.ACT_TANH, precision='bf16', **kw)
        assert torch.equal(h.t[:, :N], f32.t[:, :N].to(torch.bfloat16))
        assert torch.all(h.t[:, N:(N + 7) // 8 * 8].float() == 0)
    # an operand with the line-aligned gather pitch (what the model hands over) and a row range of a larger matrix
    big = ops.DMat.empty(M + 9, K, dev, ld=ops.gather_ld(K))
    big.t.fill_(float('nan'))                                                   # (pads beyond roundup4(K) are never read)
    big.t[:, :ops.pad4(K)] = 0
    big.t[5:5 + M, :K] = torch.from_numpy(A).to(dev)
    got = ops.gemm(big.rows(5, 5 + M), dW, precision='bf16')
    assert np.all(np.abs(got.numpy() - ref) <= env)


@pytest.mark.parametrize("M,N,K0,K1", [(1000, 600, 600, 600), (4099, 300, 300, 300), (130, 256, 256, 250), (65, 129, 300, 289), (64, 640, 590, 600),
                                       (1, 600, 600, 600), (700, 600, 600, 300), (333, 700, 600, 600)])
def test_gemm_bf16_kcat_one_launch(dev, M, N, K0, K1):
    """(round 6) dH = dZ . Wh^T + dU . Wt^T of the bf16 configuration in ONE launch of the bf16 whole-rows kernel (both reductions
    in LDS, one accumulator): against the same arithmetic in float64 -- operands rounded to bf16, exact products -- within the
    fp32 accumulation envelope; plain, accumulating, with the carry gradient G (1 - T) in the epilogue; equal to the two separate
    launches to rounding; run-to-run equal.  The last two shapes are NOT taken by the k-concatenated kernel (K paddings differ /
    N > 640): the entry point then runs the two launches itself."""
    from geographconv_amd import _ffi, ops

    def bf16(x):
        return torch.from_numpy(np.ascontiguousarray(x)).to(torch.bfloat16).double().numpy()

    A0, A1 = _rand((M, K0), 21), _rand((M, K1), 22)
    W0, W1 = _rand((N, K0), 23) * 0.3, _rand((N, K1), 24) * 0.3                 # weights as stored: transB
    d = {k: ops.DMat.from_numpy(v, dev) for k, v in dict(A0=A0, A1=A1, W0=W0, W1=W1).items()}
    ref = bf16(A0) @ bf16(W0).T + bf16(A1) @ bf16(W1).T
    env = 4e-7 * (K0 + K1) ** 0.5 * (np.abs(bf16(A0)) @ np.abs(bf16(W0)).T + np.abs(bf16(A1)) @ np.abs(bf16(W1)).T) + 1e-6
    lib = _ffi.lib()
    native = (lib.geogcn_gemm_kcat_workspace_bytes(1, M, N, K0, K1, _ffi.GEMM_BF16) > max(lib.geogcn_gemm_workspace_bytes(0, 1, M, N, K0, _ffi.GEMM_BF16),
                                                                                          lib.geogcn_gemm_workspace_bytes(0, 1, M, N, K1, _ffi.GEMM_BF16)))
    assert native == ((K0 + 31) // 32 == (K1 + 31) // 32 and N <= 640 and (K0 + 31) // 32 * 32 in (256, 320, 608))
    got = ops.gemm_kcat(d['A0'], d['W0'], d['A1'], d['W1'], transB=True, precision='bf16')
    assert np.all(np.abs(got.numpy() - ref) <= env), np.abs(got.numpy() - ref).max()
    assert torch.all(got.t[:, N:] == 0)                                         # pad columns stay zero
    assert torch.equal(got.t, ops.gemm_kcat(d['A0'], d['W0'], d['A1'], d['W1'], transB=True, precision='bf16').t)
    two = ops.gemm(d['A0'], d['W0'], transB=True, precision='bf16')
    ops.gemm(d['A1'], d['W1'], out=two, transB=True, accumulate=True, precision='bf16')
    assert np.all(np.abs(got.numpy() - two.numpy()) <= 2 * env)
    if not native:
        assert torch.equal(got.t, two.t)                                        # the entry point's own two launches
    C0 = _rand((M, N), 25)
    dC = ops.DMat.from_numpy(C0, dev)
    ops.gemm_kcat(d['A0'], d['W0'], d['A1'], d['W1'], out=dC, transB=True, accumulate=True, precision='bf16')
    assert np.all(np.abs(dC.numpy() - (ref + C0)) <= env + 1e-6)
    G, T = _rand((M, N), 26), np.random.RandomState(27).rand(M, N).astype(np.float32)
    carry = ops.GateCarry(ops.DMat.from_numpy(G, dev), ops.DMat.from_numpy(T, dev))
    fused = ops.gemm_kcat(d['A0'], d['W0'], d['A1'], d['W1'], transB=True, gate_carry=carry, precision='bf16')
    want = ref + G.astype(np.float64) * (1 - T)
    assert np.all(np.abs(fused.numpy() - want) <= env + 3e-7 * (np.abs(ref) + np.abs(G)) + 1e-6)
    # the same operands without the transposed weight layout
    d0, d1 = ops.DMat.from_numpy(np.ascontiguousarray(W0.T), dev), ops.DMat.from_numpy(np.ascontiguousarray(W1.T), dev)
    assert torch.equal(ops.gemm_kcat(d['A0'], d0, d['A1'], d1, precision='bf16').t, got.t)


@pytest.mark.parametrize("K,M,N0,N1", [(70000, 600, 600, 600), (50001, 300, 300, 256), (33333, 256, 300, 300), (20000, 600, 600, 129)])
def test_gemm_bf16_dual_tn_one_launch(dev, K, M, N0, N1):
    """(round 6) (dWh, dWt) = H^T . [dZ | dU] of the bf16 configuration in ONE launch of the bf16 A^T . B kernel (H read once): against the
    same arithmetic in float64 (operands rounded to bf16, exact products) within the fp32 accumulation envelope, equal to the two launches
    to rounding, run-to-run equal; a pair with a narrow segment (N <= 160) is not taken: the entry point runs the two launches itself."""
    from geographconv_amd import ops

    def bf16(x):
        return torch.from_numpy(np.ascontiguousarray(x)).to(torch.bfloat16).double().numpy()

    A, B0, B1 = _rand((K, M), 31), _rand((K, N0), 32), _rand((K, N1), 33)
    dA, d0, d1 = (ops.DMat.from_numpy(x, dev) for x in (A, B0, B1))
    g0, g1 = ops.gemm_dual(dA, d0, d1, transA=True, precision='bf16')
    s0 = ops.gemm(dA, d0, transA=True, precision='bf16')
    s1 = ops.gemm(dA, d1, transA=True, precision='bf16')
    for got, single, B in ((g0, s0, B0), (g1, s1, B1)):
        if min(N0, N1) > 160:          # (a narrow product stays on the exact fp32 kernel in the bf16 configuration: compared with its single launch below)
            ref = bf16(A).T @ bf16(B)
            env = 4e-7 * K ** 0.5 * (np.abs(bf16(A)).T @ np.abs(bf16(B))) + 1e-6
            assert np.all(np.abs(got.numpy() - ref) <= env), np.abs(got.numpy() - ref).max()
            assert np.all(np.abs(got.numpy() - single.numpy()) <= 2 * env)
        assert torch.all(got.t[:, B.shape[1]:] == 0)
    again = ops.gemm_dual(dA, d0, d1, transA=True, precision='bf16')
    assert torch.equal(again[0].t, g0.t) and torch.equal(again[1].t, g1.t)
    if min(N0, N1) <= 160:
        assert torch.equal(g0.t, s0.t) and torch.equal(g1.t, s1.t)


@pytest.mark.parametrize("R,M,N", [(5000, 300, 300), (4097, 300, 129), (333, 300, 600), (20000, 64, 8)])
def test_gemm_tn_splitk_matches_oracle_and_is_deterministic(dev, R, M, N):
    from geographconv_amd import ops
    A = _rand((R, M), 1)           # C = A^T . B, reduction over R rows
    B = _rand((R, N), 2)
    dA = ops.DMat.from_numpy(A, dev)
    dB = ops.DMat.from_numpy(B, dev)
    g1 = ops.gemm(dA, dB, transA=True).numpy()
    g2 = ops.gemm(dA, dB, transA=True).numpy()
    assert np.array_equal(g1, g2)
    ref = A.astype(np.float64).T @ B.astype(np.float64)
    tol = 3e-6 * (np.abs(A).T @ np.abs(B)) + 1e-5
    assert np.all(np.abs(g1 - ref) <= tol), np.abs(g1 - ref).max()


@pytest.mark.parametrize("R,M,N", [(5000, 300, 300), (4097, 300, 256), (3333, 600, 600), (1500, 129, 300), (700, 300, 129)])
def test_gemm_tn_bf16_products(dev, R, M, N):
    """bf16 configuration: dW = A^T . B with both operands rounded to bf16 (RNE) on their way into LDS, fp32
    accumulation in split-K slabs combined in fixed order: equals the exact product of the ROUNDED operands to
    fp32 accumulation error, and is run-to-run identical.  Outputs of <= 160 columns stay on the fp32 kernel."""
    from geographconv_amd import ops
    A = _rand((R, M), 1)
    B = _rand((R, N), 2)
    dA, dB = ops.DMat.from_numpy(A, dev), ops.DMat.from_numpy(B, dev)
    got = ops.gemm(dA, dB, transA=True, precision='bf16')
    Ar = torch.from_numpy(A).to(torch.bfloat16).float().numpy().astype(np.float64)
    Br = torch.from_numpy(B).to(torch.bfloat16).float().numpy().astype(np.float64)
    mag = np.abs(Ar).T @ np.abs(Br)
    if N > 160:
        assert np.all(np.abs(got.numpy() - Ar.T @ Br) <= 2e-6 * mag + 1e-6)
        assert np.abs(got.numpy() - A.astype(np.float64).T @ B.astype(np.float64)).max() > 1e-4     # really bf16
    else:
        ref = A.astype(np.float64).T @ B.astype(np.float64)
        assert np.all(np.abs(got.numpy() - ref) <= 2e-6 * (np.abs(A).T @ np.abs(B)) + 1e-6)         # exact fp32 path
    assert torch.all(got.t[:, N:] == 0)
    again = ops.gemm(dA, dB, transA=True, precision='bf16')
    assert torch.equal(got.t, again.t)
    # accumulate + bias through the ordered combine
    C0 = _rand((M, N), 3)
    dC = ops.DMat.from_numpy(C0, dev)
    ops.gemm(dA, dB, out=dC, transA=True, accumulate=True, precision='bf16')
    assert np.allclose(dC.numpy(), got.numpy() + C0, rtol=0, atol=1e-5 + 1e-6 * np.abs(got.numpy()).max())


@pytest.mark.parametrize("M,N0,N1,K", [(1000, 300, 300, 300), (4100, 600, 600, 600), (777, 129, 300, 129), (333, 64, 16, 40),
                                      (70000, 300, 300, 300)])
def test_gemm_dual_forward_equals_two_calls_bitwise(dev, M, N0, N1, K):
    """geogcn_gemm_dual_f32 (the highway block's [Wh | Wt] in one launch): every element goes through the same fp32 fma
    chain as in two geogcn_gemm_f32 calls -- the results are bitwise those, whatever pitch the outputs use."""
    from geographconv_amd import ops
    dA = ops.DMat.from_numpy(_rand((M, K), 1), dev)
    dB0, dB1 = ops.DMat.from_numpy(_rand((K, N0), 2), dev), ops.DMat.from_numpy(_rand((K, N1), 3), dev)
    b1 = torch.from_numpy(np.pad(_rand((N1,), 4), (0, ops.pad4(N1) - N1))).to(dev)
    z = ops.DMat.empty(M, N0, dev, ld=ops.gather_ld(N0))
    z.t.fill_(7.0)
    z_ref = ops.gemm(dA, dB0)
    t_ref = ops.gemm(dA, dB1, bias=b1, act=ops.ACT_SIGMOID)
    z2, t2 = ops.gemm_dual(dA, dB0, dB1, out0=z, bias1=b1, act1=ops.ACT_SIGMOID)
    assert z2 is z
    assert torch.equal(z.t[:, :ops.pad4(N0)], z_ref.t[:, :ops.pad4(N0)])      # pad columns up to roundup4 written as zeros
    assert torch.equal(t2.t, t_ref.t)
    # both linear, with and without biases
    b0 = torch.from_numpy(np.pad(_rand((N0,), 5), (0, ops.pad4(N0) - N0))).to(dev)
    c0, c1 = ops.gemm_dual(dA, dB0, dB1, bias0=b0, act0=ops.ACT_TANH, act1=ops.ACT_TANH)
    assert torch.equal(c0.t, ops.gemm(dA, dB0, bias=b0, act=ops.ACT_TANH).t)
    assert torch.equal(c1.t, ops.gemm(dA, dB1, act=ops.ACT_TANH).t)
    with pytest.raises(Exception):
        ops.gemm_dual(dA, dB0, dB1, act0=ops.ACT_TANH, act1=ops.ACT_SIGMOID)


@pytest.mark.parametrize("R,M,N0,N1", [(5000, 300, 300, 300), (4097, 300, 256, 300), (3333, 600, 600, 600), (20000, 64, 8, 40),
                                      (0, 300, 300, 300)])
def test_gemm_dual_transposed_equals_two_calls_bitwise(dev, R, M, N0, N1):
    """(dWh, dWt) = H^T . [dZ | dU] in one pass over H: same split-K slices and the same ordered slab combine as two
    calls => bitwise equal, deterministic; an empty reduction (a rank without rows) gives zeros."""
    from geographconv_amd import ops
    dA = ops.DMat.from_numpy(_rand((R, M), 1), dev) if R else ops.DMat(0, M, dev)
    dB0 = ops.DMat.from_numpy(_rand((R, N0), 2), dev) if R else ops.DMat(0, N0, dev)
    dB1 = ops.DMat.from_numpy(_rand((R, N1), 3), dev) if R else ops.DMat(0, N1, dev)
    o0, o1 = ops.DMat(M, N0, dev), ops.DMat(M, N1, dev)
    o0.t.fill_(3.0)
    o1.t.fill_(3.0)
    ops.gemm_dual(dA, dB0, dB1, out0=o0, out1=o1, transA=True)
    if R == 0:
        assert torch.all(o0.t[:, :N0] == 0) and torch.all(o1.t[:, :N1] == 0)
        return
    r0, r1 = ops.gemm(dA, dB0, transA=True), ops.gemm(dA, dB1, transA=True)
    # the split-K plan of the dual launch covers twice the tiles: slices may differ from the single call's, so
    # compare within the fp32 envelope, and bitwise against a second dual launch
    A, B0, B1 = dA.numpy().astype(np.float64), dB0.numpy().astype(np.float64), dB1.numpy().astype(np.float64)
    for got, B in ((o0, B0), (o1, B1)):
        ref = A.T @ B
        assert np.all(np.abs(got.numpy() - ref) <= 3e-6 * (np.abs(A).T @ np.abs(B)) + 1e-5)
    for got, single in ((o0, r0), (o1, r1)):
        assert np.abs(got.numpy() - single.numpy()).max() <= 2e-6 * np.abs(single.numpy()).max() * np.sqrt(R) + 1e-5
    p0, p1 = ops.gemm_dual(dA, dB0, dB1, transA=True)
    assert torch.equal(p0.t, o0.t) and torch.equal(p1.t, o1.t)
    assert torch.all(o0.t[:, N0:] == 0) and torch.all(o1.t[:, N1:] == 0)


@pytest.mark.parametrize("R,M,N", [(9475, 300, 300), (50001, 300, 256), (4099, 256, 300), (9475, 129, 161), (20003, 257, 320),
                                   (1037, 300, 300), (130, 300, 300), (440000 // 8 + 3, 384, 300)])
def test_gemm_tn_without_lds_tiles_pitches_and_slab_edges(dev, R, M, N):
    """gemm_tn_direct_kernel (A^T . B with every fragment loaded straight into registers: the four 8-wave tile shapes 160 / 128 x
    320 / 256, reduction lengths that are not multiples of 4 or of the slab length, outputs that do not fill the last 16 x 16
    tile, operands that are ROW RANGES of pitched matrices with garbage beyond their columns and rows -- the kernel masks by
    column and bounds each slab's descriptor, so nothing outside may leak in), against the fp64 product; run to run bitwise."""
    from geographconv_amd import ops
    A = _rand((R, M), 11)
    B = _rand((R, N), 12)
    ref = A.astype(np.float64).T @ B.astype(np.float64)
    tol = 3e-6 * (np.abs(A).T @ np.abs(B)) + 1e-5
    # operands embedded in larger buffers: pitch beyond the width, rows before and after, everything outside = NaN-free garbage
    bigA = ops.DMat.empty(R + 7, M, dev, ld=ops.gather_ld(M) + 8)
    bigB = ops.DMat.empty(R + 7, N, dev, ld=ops.gather_ld(N))
    bigA.t.fill_(1e30)
    bigB.t.fill_(-1e30)
    bigA.t[3:3 + R, :M] = torch.from_numpy(A).to(dev)
    bigB.t[3:3 + R, :N] = torch.from_numpy(B).to(dev)
    # (the library's convention: pad columns up to roundup4 are zero)
    bigA.t[3:3 + R, M:ops.pad4(M)] = 0
    bigB.t[3:3 + R, N:ops.pad4(N)] = 0
    dA, dB = bigA.rows(3, 3 + R), bigB.rows(3, 3 + R)
    g1 = ops.gemm(dA, dB, transA=True)
    assert np.all(np.abs(g1.numpy() - ref) <= tol), np.abs(g1.numpy() - ref).max()
    assert torch.all(g1.t[:, N:] == 0)
    g2 = ops.gemm(dA, dB, transA=True)
    assert torch.equal(g1.t, g2.t)
    # accumulate + bias + activation go through the ordered combine
    bias = torch.from_numpy(_rand((ops.pad4(N),), 13)).to(dev)
    C0 = _rand((M, N), 14)
    dC = ops.DMat.from_numpy(C0, dev)
    ops.gemm(dA, dB, out=dC, transA=True, bias=bias, act=ops.ACT_TANH, accumulate=True)
    want = np.tanh(ref + bias.cpu().numpy()[:N].astype(np.float64)) + C0
    assert np.all(np.abs(dC.numpy() - want) <= tol + 2e-6)


@pytest.mark.parametrize("M,K,N0,N1", [(40000, 300, 300, 300), (33001, 300, 300, 600), (32768, 256, 300, 300), (50001, 290, 620, 289),
                                      (40001, 300, 256, 512), (33000, 256, 250, 500)])
def test_whole_rows_kernel_equals_the_staged_kernel_bitwise(dev, M, K, N0, N1):
    """The fused highway launches on gemm_rows_kernel (64 whole rows of A per block, weights in fragment order; taken when the
    caller supplies the workspace): the dual launch and the k-concatenated one are BIT-identical to the staged kernel -- the
    same calls with a NULL workspace -- through bias, sigmoid, accumulate, both weight layouts, gather-pitch outputs and a row
    count that is not a multiple of 64."""
    import ctypes as C
    from geographconv_amd import _ffi, ops
    lib = _ffi.lib()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    A = ops.DMat.from_numpy(_rand((M, K), 1), dev)
    W0, W1 = ops.DMat.from_numpy(_rand((K, N0), 2, 0.1), dev), ops.DMat.from_numpy(_rand((K, N1), 3, 0.1), dev)
    b1 = torch.from_numpy(_rand((ops.pad4(N1),), 4)).to(dev)
    assert lib.geogcn_gemm_dual_workspace_bytes(0, M, N0, N1, K, _ffi.GEMM_F32) > 0            # this shape is taken by the rows kernel
    got0 = ops.DMat.empty(M, N0, dev, ld=ops.gather_ld(N0))
    got0.t.zero_()
    got0, got1 = ops.gemm_dual(A, W0, W1, out0=got0, bias1=b1, act1=ops.ACT_SIGMOID)
    ref0 = ops.DMat.empty(M, N0, dev, ld=ops.gather_ld(N0))
    ref0.t.zero_()
    ref1 = ops.DMat(M, N1, dev)
    _ffi.check(lib.geogcn_gemm_dual_f32(0, M, N0, N1, K, ops._p(A.t), A.ld, ops._p(W0.t), W0.ld, ops._p(W1.t), W1.ld,
                                        ops._p(ref0.t), ref0.ld, ops._p(ref1.t), ref1.ld, None, 0, ops._p(b1), ops.ACT_SIGMOID,
                                        _ffi.GEMM_F32, None, 0, st), 'dual, staged')
    assert torch.equal(got0.t, ref0.t) and torch.equal(got1.t, ref1.t)
    assert torch.all(got1.t[:, N1:] == 0)
    # and against the fp64 product
    ref = A.numpy().astype(np.float64) @ W0.numpy().astype(np.float64)
    assert np.all(np.abs(got0.numpy() - ref) <= 2e-6 * (np.abs(A.numpy()) @ np.abs(W0.numpy())) + 1e-6)
    # k-concatenated: dH = dZ . Wh^T + dU . Wt^T [+ carry], weights as stored (N x K) and transposed
    if N0 <= 320:
        G0, G1 = A, ops.DMat.from_numpy(_rand((M, K), 5), dev)
        for transB in (True, False):
            V0 = ops.DMat.from_numpy(_rand((N0, K) if transB else (K, N0), 6, 0.1), dev)
            V1 = ops.DMat.from_numpy(_rand((N0, K) if transB else (K, N0), 7, 0.1), dev)
            assert lib.geogcn_gemm_kcat_workspace_bytes(int(transB), M, N0, K, K, _ffi.GEMM_F32) > 0
            for acc in (False, True):
                carry = _rand((M, N0), 8)
                out = ops.DMat.from_numpy(carry, dev)
                want = ops.DMat.from_numpy(carry, dev)
                ops.gemm_kcat(G0, V0, G1, V1, out=out, transB=transB, accumulate=acc)
                _ffi.check(lib.geogcn_gemm_kcat_f32(int(transB), M, N0, K, K, ops._p(G0.t), G0.ld, ops._p(V0.t), V0.ld, ops._p(G1.t),
                                                    G1.ld, ops._p(V1.t), V1.ld, ops._p(want.t), want.ld, int(acc), _ffi.GEMM_F32, None, 0, st),
                           'kcat, staged')
                assert torch.equal(out.t, want.t), (transB, acc)


@pytest.mark.parametrize("M,N,K,transB", [(40000, 300, 256, True), (33001, 300, 300, True), (50001, 289, 250, True), (32768, 600, 300, True),
                                         (40000, 256, 300, False), (33001, 250, 256, False), (32768, 512, 300, False), (40000, 256, 300, True)])
def test_single_products_on_the_whole_rows_kernel_bitwise(dev, M, N, K, transB):
    """dH = dS . W^T (the output layer's and plain layers' backward) and A . B products of 256 / 512 columns (the logits: four waves
    x four column tiles) take gemm_rows_kernel when the caller supplies the workspace geogcn_gemm_workspace_bytes asks for:
    BIT-identical to the staged kernel (same call, NULL workspace) through bias, activation and accumulate; a 300-wide A . B
    stays on the staged kernel (no workspace asked)."""
    import ctypes as C
    from geographconv_amd import _ffi, ops
    lib = _ffi.lib()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert lib.geogcn_gemm_workspace_bytes(0, int(transB), M, N, K, _ffi.GEMM_F32) > 0
    assert lib.geogcn_gemm_workspace_bytes(0, 0, M, 300, K, _ffi.GEMM_F32) == 0
    assert lib.geogcn_gemm_workspace_bytes(0, int(transB), 1000, N, K, _ffi.GEMM_F32) == 0          # too few rows: staged
    A = ops.DMat.from_numpy(_rand((M, K), 1), dev)
    W = ops.DMat.from_numpy(_rand((N, K) if transB else (K, N), 2, 0.1), dev)
    bias = torch.from_numpy(_rand((ops.pad4(N),), 3)).to(dev)
    for acc, b, act in ((False, None, ops.ACT_NONE), (True, None, ops.ACT_NONE), (False, bias, ops.ACT_TANH)):
        C0 = _rand((M, N), 4)
        got, want = ops.DMat.from_numpy(C0, dev), ops.DMat.from_numpy(C0, dev)
        ops.gemm(A, W, out=got, transB=transB, bias=b, act=act, accumulate=acc)
        _ffi.check(lib.geogcn_gemm_f32(0, int(transB), M, N, K, ops._p(A.t), A.ld, ops._p(W.t), W.ld, ops._p(want.t), want.ld, ops._p(b),
                                       act, int(acc), _ffi.GEMM_F32, None, 0, st), 'single product, staged')
        assert torch.equal(got.t, want.t), (acc, act)
    Wn = W.numpy().astype(np.float64)
    Wn = Wn.T if transB else Wn
    ref = A.numpy().astype(np.float64) @ Wn
    got = ops.gemm(A, W, transB=transB)
    assert np.all(np.abs(got.numpy() - ref) <= 2e-6 * (np.abs(A.numpy()) @ np.abs(Wn)) + 1e-6)
    # a gather-pitch output keeps its pad columns
    wide = ops.DMat.empty(M, N, dev, ld=ops.gather_ld(N) + 32)
    wide.t.fill_(7.0)
    ops.gemm(A, W, out=wide, transB=transB)
    assert torch.equal(wide.t[:, :N], got.t[:, :N]) and torch.all(wide.t[:, ops.pad4(N):] == 7.0)


@pytest.mark.parametrize("M,K,N0,N1", [(4100, 600, 600, 600), (3001, 300, 300, 300), (2000, 256, 300, 256), (1500, 600, 289, 620), (700, 129, 300, 300),
                                      (900, 300, 700, 300)])
def test_gemm_dual_bf16_equals_the_two_launches_bitwise(dev, M, K, N0, N1):
    """geogcn_gemm_dual_bf16 (the bf16 configuration's H . [Wh | Wt]: one launch of the bf16 whole-rows kernel with two column
    segments, or the two launches where a width does not fit): Z as bf16 / fp32 and T = sigmoid(H . Wt + bt) are BIT-identical to
    geogcn_gemm_f32_bf16c / geogcn_gemm_f32 with precision bf16."""
    from geographconv_amd import ops
    A = ops.DMat.from_numpy(_rand((M, K), 1), dev)
    W0, W1 = ops.DMat.from_numpy(_rand((K, N0), 2, 0.1), dev), ops.DMat.from_numpy(_rand((K, N1), 3, 0.1), dev)
    b1 = torch.from_numpy(_rand((ops.pad4(N1),), 4)).to(dev)
    z_ref = ops.gemm(A, W0, out=ops.HMat(M, N0, dev), precision='bf16')
    t_ref = ops.gemm(A, W1, bias=b1, act=ops.ACT_SIGMOID, precision='bf16')
    z, t = ops.gemm_dual_bf16(A, W0, W1, bias1=b1, act1=ops.ACT_SIGMOID)
    assert isinstance(z, ops.HMat) and torch.equal(z.t.view(torch.int16)[:, :N0], z_ref.t.view(torch.int16)[:, :N0])
    assert torch.all(z.t[:, N0:(N0 + 7) // 8 * 8] == 0) and torch.equal(t.t, t_ref.t)
    z32_ref = ops.gemm(A, W0, precision='bf16')
    z32, t2 = ops.gemm_dual_bf16(A, W0, W1, out0=ops.DMat.empty(M, N0, dev), bias1=b1, act1=ops.ACT_SIGMOID)
    assert torch.equal(z32.t, z32_ref.t) and torch.equal(t2.t, t_ref.t)
    ref = A.numpy().astype(np.float64) @ W0.numpy().astype(np.float64)
    assert np.abs(z32.numpy() - ref).max() <= 2e-2 * np.abs(ref).max()


@pytest.mark.parametrize("M,F,pitched", [(40000, 300, False), (33001, 256, True), (50001, 289, True), (5000, 300, False), (777, 129, True),
                                         (4100, 600, False), (3000, 620, True)])
def test_gemm_kcat_with_the_carry_gradient_in_its_epilogue_bitwise(dev, M, F, pitched):
    """geogcn_gemm_kcat_gated_f32: dH = dZ . Wh^T + dU . Wt^T + G * (1 - T) with the highway block's carry formed in the epilogue
    of the whole-rows kernel (large M) or by geogcn_gate_carry_f32 ahead of the accumulating call (any other shape) is BIT-identical
    to highway_bwd's stored carry + the accumulating call; highway_bwd without its carry output leaves dS, dU and the two bias
    gradients unchanged."""
    from geographconv_amd import ops
    ld = ops.gather_ld(F) if pitched else None
    G = ops.DMat.empty(M, F, dev, ld=ld)
    G.t.zero_()
    G.t[:, :F].copy_(torch.from_numpy(_rand((M, F), 1)))
    T = ops.DMat.from_numpy(1.0 / (1.0 + np.exp(-3.0 * _rand((M, F), 2))), dev)
    Hc, H = ops.DMat.from_numpy(np.tanh(_rand((M, F), 3)), dev), ops.DMat.from_numpy(_rand((M, F), 4), dev)
    if G.ld != T.ld:                 # (highway_bwd takes ONE pitch for its four inputs)
        Gp = ops.DMat.from_numpy(G.numpy(), dev)
    else:
        Gp = G
    b = [torch.zeros(ops.pad4(F), device=dev) for _ in range(4)]
    dS, dU, carry = ops.highway_bwd(Gp, T, Hc, H, dbS=b[0], dbU=b[1])
    dS2, dU2, none = ops.highway_bwd(Gp, T, Hc, H, dbS=b[2], dbU=b[3], carry=False)
    assert none is None and torch.equal(dS.t[:, :F], dS2.t[:, :F]) and torch.equal(dU.t[:, :F], dU2.t[:, :F])     # (dS has the gather pitch)
    assert torch.equal(b[0], b[2]) and torch.equal(b[1], b[3])
    assert torch.equal(ops.GateCarry(G, T).dense().t[:, :F], carry.t[:, :F])
    dZ = ops.DMat.from_numpy(_rand((M, F), 5), dev)
    for transB in (True, False):
        Wh, Wt = ops.DMat.from_numpy(_rand((F, F), 6, 0.1), dev), ops.DMat.from_numpy(_rand((F, F), 7, 0.1), dev)
        want = ops.DMat.from_numpy(carry.numpy(), dev)
        ops.gemm_kcat(dZ, Wh, dU, Wt, out=want, transB=transB, accumulate=True)
        got = ops.gemm_kcat(dZ, Wh, dU, Wt, transB=transB, gate_carry=ops.GateCarry(G, T))
        assert torch.equal(got.t, want.t), transB
        wide = ops.DMat.empty(M, F, dev, ld=ops.gather_ld(F) + 32)
        wide.t.fill_(7.0)
        ops.gemm_kcat(dZ, Wh, dU, Wt, out=wide, transB=transB, gate_carry=ops.GateCarry(G, T))
        assert torch.equal(wide.t[:, :F], want.t[:, :F]) and torch.all(wide.t[:, ops.gather_ld(F):] == 7.0)
    with pytest.raises(ValueError):
        ops.gemm_kcat(dZ, Wh, dU, Wt, out=want, transB=True, accumulate=True, gate_carry=ops.GateCarry(G, T))
    # ... and with the dropout + tanh gradient of the layer below in the same epilogue (geogcn_gemm_kcat_gated_tanhbwd_f32), against
    # the gated product followed by geogcn_act_bwd_f32 with the mask; the bias gradient: the column sums, against the fused pass
    if F % 4 == 0:
        Y0 = ops.DMat.from_numpy(np.tanh(_rand((M, F), 21)), dev)
        keep = torch.from_numpy((np.random.RandomState(22).rand(M, F) < 0.5).astype(np.uint8)).to(dev)
        for transB in (True, False):
            Wh, Wt = ops.DMat.from_numpy(_rand((F, F), 6, 0.1), dev), ops.DMat.from_numpy(_rand((F, F), 7, 0.1), dev)
            dH = ops.gemm_kcat(dZ, Wh, dU, Wt, transB=transB, gate_carry=ops.GateCarry(G, T))
            db_ref = torch.zeros(ops.pad4(F), device=dev)
            want = ops.act_bwd_colsum(dH, Y0, ops.ACT_TANH, db_ref, out=ops.DMat.empty(M, F, dev, ld=ops.gather_ld(F)), keep_mask=keep, scale=2.0)
            got = ops.DMat.empty(M, F, dev, ld=ops.gather_ld(F))
            got.t.fill_(7.0)
            ops.gemm_kcat(dZ, Wh, dU, Wt, out=got, transB=transB, gate_carry=ops.GateCarry(G, T), tanh_bwd=(Y0, keep, 2.0))
            assert torch.equal(got.t[:, :ops.pad4(F)], want.t[:, :ops.pad4(F)]), transB
            assert torch.equal(ops.colsum_rowblocks(got, torch.zeros(ops.pad4(F), device=dev)), db_ref)
    # ONE product with the carry in its epilogue (geogcn_gemm_gated_f32: the bf16 configuration's first of two launches; exact
    # fp32 and bf16x3 as well), against the stored carry + the accumulating call of the same precision
    for prec in ('bf16', 'f32', 'bf16x3'):
        for transB in (True, False):
            want = ops.DMat.from_numpy(carry.numpy(), dev)
            ops.gemm(dZ, Wh, out=want, transB=transB, accumulate=True, precision=prec)
            got = ops.gemm(dZ, Wh, transB=transB, precision=prec, gate_carry=ops.GateCarry(G, T))
            assert torch.equal(got.t, want.t), (prec, transB)
    with pytest.raises(ValueError):
        ops.gemm(dZ, Wh, transB=True, accumulate=True, out=want, gate_carry=ops.GateCarry(G, T))


@pytest.mark.parametrize("M,N,K0,K1", [(1000, 300, 300, 300), (4100, 600, 600, 600), (777, 300, 129, 300), (333, 16, 40, 64),
                                      (70000, 300, 300, 300)])
def test_gemm_kcat_two_products_one_accumulator(dev, M, N, K0, K1):
    """dH = dZ . Wh^T + dU . Wt^T [+ carry] as ONE contraction over K0 + K1 (geogcn_gemm_kcat_f32)."""
    from geographconv_amd import ops
    A0, A1 = _rand((M, K0), 1), _rand((M, K1), 2)
    B0, B1 = _rand((N, K0), 3), _rand((N, K1), 4)           # weights as stored: n_in x n_out = N x K
    C0 = _rand((M, N), 5)
    dA0, dA1 = ops.DMat.from_numpy(A0, dev), ops.DMat.from_numpy(A1, dev)
    dB0, dB1 = ops.DMat.from_numpy(B0, dev), ops.DMat.from_numpy(B1, dev)
    ref = A0.astype(np.float64) @ B0.astype(np.float64).T + A1.astype(np.float64) @ B1.astype(np.float64).T
    tol = 2e-6 * (np.abs(A0) @ np.abs(B0).T + np.abs(A1) @ np.abs(B1).T) + 1e-6
    got = ops.gemm_kcat(dA0, dB0, dA1, dB1, transB=True)
    assert np.all(np.abs(got.numpy() - ref) <= tol)
    assert torch.all(got.t[:, N:] == 0)
    dC = ops.DMat.from_numpy(C0, dev)
    ops.gemm_kcat(dA0, dB0, dA1, dB1, out=dC, transB=True, accumulate=True)
    assert np.all(np.abs(dC.numpy() - (ref + C0)) <= tol + 1e-6)
    assert torch.equal(ops.gemm_kcat(dA0, dB0, dA1, dB1, transB=True).t, got.t)           # deterministic
    # NN form
    dB0n, dB1n = ops.DMat.from_numpy(np.ascontiguousarray(B0.T), dev), ops.DMat.from_numpy(np.ascontiguousarray(B1.T), dev)
    got = ops.gemm_kcat(dA0, dB0n, dA1, dB1n)
    assert np.all(np.abs(got.numpy() - ref) <= tol)


def _bow(n_docs, n_words, mean, seed):
    X = synth.bow_x(n_docs, n_words, mean, seed=seed)
    return X


@pytest.mark.parametrize("n_docs,n_words,mean,F", [(30000, 2000, 40, 300), (9000, 700, 25, 129), (5000, 300, 12, 64),
                                                  (20000, 1500, 30, 600), (700, 90, 6, 5), (2049, 50, 4, 1024)])
def test_xt_dot_document_blocked_matches_oracle(dev, n_docs, n_words, mean, F, monkeypatch):
    """geogcn_xt_dot_f32 (dW0 = X^T . dS0, reference gcnmodel.py:39 autodiff): the document-blocked kernel against the
    fp64 product, against the plain row-gather SpMM on CSR(X^T), and against itself (bitwise reproducible); rows of
    X^T without nonzeros come out as zeros; value-dropout variants reuse the plan."""
    from geographconv_amd import ops
    monkeypatch.setattr(tuning, 'XT_MIN_NNZ', 0)
    X = _bow(n_docs, n_words, mean, seed=F)
    X = sps.csr_matrix(X)
    X.data[::7] *= -1.0                                    # signs: catches an absolute-value / ordering slip
    x = ops.SparseOperand.from_scipy(X, dev, dense_head=False)
    G = _rand((n_docs, F), 3)
    dG = ops.DMat.from_numpy(G, dev)
    assert x.xt_plan(F) is not None
    out = ops.DMat(n_words, F, dev)
    out.t.fill_(5.0)
    ops.spmm_t(x, dG, out=out)
    ref = (X.T.astype(np.float64) @ G.astype(np.float64))
    mag = np.asarray(abs(X).T.astype(np.float64) @ np.abs(G).astype(np.float64))
    assert np.all(np.abs(out.numpy() - ref) <= 2e-6 * mag + 1e-6), np.abs(out.numpy() - ref).max()
    assert torch.all(out.t[:, F:] == 0)
    empty = np.diff(sps.csr_matrix(X.T).indptr) == 0
    if empty.any():
        assert np.all(out.numpy()[empty] == 0)
    again = ops.spmm_t(x, dG)
    assert torch.equal(again.t[:, :F], out.t[:, :F])
    plain = ops.spmm(x.bwd, dG)                              # the row gather: same numbers up to summation order
    assert np.all(np.abs(plain.numpy() - out.numpy()) <= 4e-6 * mag + 1e-6)
    # G with the line-aligned pitch
    dG2 = ops.DMat.empty(n_docs, F, dev, ld=ops.gather_ld(F))
    dG2.copy_from(dG)
    assert torch.equal(ops.spmm_t(x, dG2).t[:, :F], out.t[:, :F])
    # wide outputs run as column slabs of <= XT_MAX_F; in one piece the rows are cut into different parts (the plan
    # depends on the width), so: the same numbers up to summation order
    monkeypatch.setattr(tuning, 'XT_MAX_F', 1 << 30)
    assert np.all(np.abs(ops.spmm_t(x, dG).numpy() - out.numpy()) <= 4e-6 * mag + 1e-6)


def test_xt_dot_with_dense_head_and_value_dropout(dev, monkeypatch):
    """The split transpose (dense head panel on the MFMA pipe + document-blocked tail) and its value-dropout variant
    (SparseInputDropoutLayer: same structure, same plan, new values) against the fp64 products."""
    from geographconv_amd import ops
    monkeypatch.setattr(tuning, 'XT_MIN_NNZ', 0)
    X = sps.csr_matrix(_bow(20000, 1500, 30, seed=1))
    x = ops.SparseOperand.from_scipy(X, dev)
    assert x.head_dense is not None and x.xt_plan(300) is not None
    G = _rand((20000, 300), 3)
    dG = ops.DMat.from_numpy(G, dev)
    got = ops.spmm_t(x, dG).numpy()
    ref = X.T.astype(np.float64) @ G.astype(np.float64)
    mag = np.asarray(abs(X).T.astype(np.float64) @ np.abs(G).astype(np.float64))
    assert np.all(np.abs(got - ref) <= 3e-6 * mag + 1e-5)
    xd = ops.sparse_dropout(x, 0.4, seed=11, call=2)
    assert xd.xt_plan(300) is x.xt_plan(300)
    Xd = sps.csr_matrix((xd.fwd.val.cpu().numpy(), X.indices, X.indptr), shape=X.shape)
    got = ops.spmm_t(xd, dG).numpy()
    ref = Xd.T.astype(np.float64) @ G.astype(np.float64)
    assert np.all(np.abs(got - ref) <= 3e-6 * mag / 0.6 + 1e-5)
    # forward as well: X . W with the dropped values
    W = _rand((1500, 300), 4, 0.1)
    dWm = ops.DMat.from_numpy(W, dev)
    got = ops.spmm_x(xd, dWm).numpy()
    ref = Xd.astype(np.float64) @ W.astype(np.float64)
    assert np.all(np.abs(got - ref) <= 3e-6 * np.asarray(abs(Xd).astype(np.float64) @ np.abs(W)) + 1e-6)


@pytest.mark.parametrize("n_docs,n_words,mean,F", [(30000, 2000, 40, 300), (9000, 700, 25, 129), (4000, 600, 30, 600)])
def test_spmm_x_below_the_hot_threshold_and_the_accumulate_form(dev, n_docs, n_words, mean, F):
    """X . W0 + b0 with tanh (reference gcnmodel.py:39-42) for a small X (plain row gather: below tuning.HOT_MIN_NNZ)
    against the fp64 product; deterministic.  And the accumulate form C = C0 + A.B (geogcn_spmm_csr_acc_f32) with long
    rows through the chunk path."""
    from geographconv_amd import ops
    X = sps.csr_matrix(_bow(n_docs, n_words, mean, seed=2))
    x = ops.SparseOperand.from_scipy(X, dev)
    assert x.head_dense is not None
    W = _rand((n_words, F), 4, 0.1)
    b = _rand((F,), 5, 0.1)
    dWm = ops.DMat.from_numpy(W, dev)
    db = torch.from_numpy(np.pad(b, (0, ops.pad4(F) - F))).to(dev)
    got = ops.spmm_x(x, dWm, bias=db, act=ops.ACT_TANH)
    ref = np.tanh(X.astype(np.float64) @ W.astype(np.float64) + b)
    tol = 3e-6 * np.asarray(abs(X).astype(np.float64) @ np.abs(W)) + 1e-6
    assert np.all(np.abs(got.numpy() - ref) <= tol)
    assert torch.all(got.t[:, F:] == 0)
    assert torch.equal(ops.spmm_x(x, dWm, bias=db, act=ops.ACT_TANH).t, got.t)
    A = _skewed_csr(600, n_words, seed=9)
    C0 = _rand((600, F), 6)
    dC = ops.DMat.from_numpy(C0, dev)
    ops.spmm(ops.CSR(A, dev), dWm, out=dC, accumulate=True)
    ref = C0 + A.astype(np.float64) @ W.astype(np.float64)
    assert np.all(np.abs(dC.numpy() - ref) <= 3e-6 * (np.abs(C0) + np.asarray(abs(A).astype(np.float64) @ np.abs(W))) + 1e-6)


@pytest.mark.parametrize("n_docs,n_words,mean,F,n_hot", [(30000, 2000, 40, 300, None), (9000, 700, 25, 129, None), (5000, 300, 12, 64, 7),
                                                        (4000, 600, 30, 384, None), (3000, 50, 6, 5, 0), (2500, 90, 6, 20, 1000)])
def test_spmm_hot_rows_in_lds(dev, n_docs, n_words, mean, F, n_hot):
    """geogcn_spmm_csr_hot_f32 (X . W0 with the hot rows of W0 in LDS): against the fp64 product and the plain gather
    kernel; the [hot | cold] reordering only changes the summation order; bitwise reproducible; every n_hot from 0 to
    "all columns" is valid."""
    from geographconv_amd import _ffi, ops
    X = sps.csr_matrix(_bow(n_docs, n_words, mean, seed=3))
    X.data[::5] *= -1.0
    csr = ops.CSR(X, dev)
    cap = int(_ffi.lib().geogcn_spmm_hot_capacity(F))
    assert cap > 0
    hot = ops.HotCSR(csr, X.data, cap if n_hot is None else min(n_hot, cap))
    W = _rand((n_words, F), 4, 0.1)
    b = _rand((F,), 5, 0.1)
    dWm = ops.DMat.from_numpy(W, dev)
    db = torch.from_numpy(np.pad(b, (0, ops.pad4(F) - F))).to(dev)
    got = ops.DMat(n_docs, F, dev)
    got.t.fill_(9.0)
    ops.spmm_hot(hot, dWm, out=got, bias=db, act=ops.ACT_TANH)
    ref = np.tanh(X.astype(np.float64) @ W.astype(np.float64) + b)
    tol = 3e-6 * np.asarray(abs(X).astype(np.float64) @ np.abs(W)) + 1e-6
    assert np.all(np.abs(got.numpy() - ref) <= tol)
    assert torch.all(got.t[:, F:] == 0)
    assert torch.equal(ops.spmm_hot(hot, dWm, bias=db, act=ops.ACT_TANH).t, got.t)
    plain = ops.spmm(csr, dWm, bias=db, act=ops.ACT_TANH)
    assert np.all(np.abs(plain.numpy() - got.numpy()) <= 2 * tol)
    if n_hot == 0:
        assert hot.hot_fraction == 0.0
        assert torch.equal(ops.spmm_hot(hot, dWm).t, ops.spmm(csr, dWm).t)        # no hot part: the same order, bitwise
    assert int(_ffi.lib().geogcn_spmm_hot_capacity(2000)) == 0


@pytest.mark.parametrize("F", [600, 768, 392])
def test_spmm_x_wide_layer_runs_as_two_column_slabs_of_the_lds_kernel(dev, F, monkeypatch):
    """A first layer wider than the LDS kernel's 384 columns (configs[4]: 600): ops.spmm_x runs the kernel on two column
    slabs of W0 / bias / output -- each slab bitwise what the kernel gives on a copy of those columns -- instead of falling
    back to the plain row gather."""
    from geographconv_amd import ops, tuning
    monkeypatch.setattr(tuning, 'HOT_MIN_NNZ', 0)
    X = sps.csr_matrix(_bow(6000, 900, 30, seed=3))
    x = ops.SparseOperand.from_scipy(X, dev)
    W = _rand((900, F), 4, 0.1)
    b = _rand((F,), 5, 0.1)
    dWm = ops.DMat.from_numpy(W, dev)
    db = torch.from_numpy(b).to(dev)
    got = ops.spmm_x(x, dWm, bias=db, act=ops.ACT_TANH)
    ref = np.tanh(X.astype(np.float64) @ W.astype(np.float64) + b)
    tol = 3e-6 * np.asarray(abs(X).astype(np.float64) @ np.abs(W)) + 1e-6
    assert np.all(np.abs(got.numpy() - ref) <= tol)
    half = F // 2
    (cap, hot), = x._hot.items()                                   # ONE reordered copy of X, sized for a slab
    assert hot.n_hot > 0 and hot.n_hot <= cap
    for s in range(2):
        Ws = ops.DMat.from_numpy(W[:, s * half:(s + 1) * half], dev)
        bs = torch.from_numpy(b[s * half:(s + 1) * half].copy()).to(dev)
        one = ops.spmm_hot(hot, Ws, bias=bs, act=ops.ACT_TANH)
        assert torch.equal(one.t[:, :half], got.t[:, s * half:(s + 1) * half])


@pytest.mark.parametrize("n_docs,n_words,mean,F,p", [(20000, 1500, 30, 300, 0.5), (7001, 700, 25, 64, 0.2), (3000, 300, 12, 128, 0.8)])
def test_spmm_hot_with_the_dropout_in_its_epilogue(dev, n_docs, n_words, mean, F, p, monkeypatch):
    """geogcn_spmm_csr_hot_dropout_f32 == geogcn_spmm_csr_hot_f32 + geogcn_dropout_mask_philox + geogcn_dropout_apply_f32,
    bit for bit (activation, dropped copy, mask bytes): Philox stream at a non-zero offset, the device-counter form of a
    captured step, and an injected mask.  F % 4 != 0 is refused (the caller then runs the separate kernels)."""
    from geographconv_amd import _ffi, ops
    monkeypatch.setattr(tuning, 'HOT_MIN_NNZ', 0)
    X = sps.csr_matrix(_bow(n_docs, n_words, mean, seed=5))
    x = ops.SparseOperand.from_scipy(X, dev)
    W = ops.DMat.from_numpy(_rand((n_words, F), 4, 0.1), dev)
    b = torch.from_numpy(_rand((F,), 5, 0.1)).to(dev)
    seed, offset = 123456789, 77 * n_docs * F // 4
    H0 = ops.spmm_x(x, W, bias=b, act=ops.ACT_TANH)
    mask = ops.dropout_mask(n_docs, F, p, seed, offset, dev)
    Hd = ops.dropout_apply(H0, mask, p)
    got = ops.spmm_x_dropout(x, W, b, ops.ACT_TANH, p, seed=seed, offset=offset)
    assert got is not None
    assert torch.equal(got[0].t, H0.t) and torch.equal(got[2], mask) and torch.equal(got[1].t, Hd.t)
    assert 0.9 * (1 - p) < float(mask.float().mean()) < 1.1 * (1 - p)
    # device-resident call counter (hipGraph replays): offset = (calls * per_call + base) / 4
    calls = torch.tensor([77], dtype=torch.int64, device=dev)
    got = ops.spmm_x_dropout(x, W, b, ops.ACT_TANH, p, seed=seed, calls_dev=calls, per_call=n_docs * F, base=0)
    assert torch.equal(got[2], mask) and torch.equal(got[1].t, Hd.t)
    # injected mask
    inj = (torch.rand((n_docs, F), device=dev) < 0.5).to(torch.uint8)
    got = ops.spmm_x_dropout(x, W, b, ops.ACT_TANH, p, mask_in=inj)
    assert got[2] is inj and torch.equal(got[0].t, H0.t) and torch.equal(got[1].t, ops.dropout_apply(H0, inj, p).t)
    # not applicable: width not a multiple of 4, p = 0
    W7 = ops.DMat.from_numpy(_rand((n_words, 30), 4, 0.1), dev)
    assert ops.spmm_x_dropout(x, W7, None, ops.ACT_TANH, p) is None and ops.spmm_x_dropout(x, W, b, ops.ACT_TANH, 0.0) is None
    lib = _ffi.lib()
    hot = next(iter(x._hot.values()))
    rc = lib.geogcn_spmm_csr_hot_dropout_f32(n_docs, n_words, ops._p(hot.rowptr), ops._p(hot.rowsplit), ops._p(hot.colidx), ops._p(hot.val),
                                             ops._p(W7.t), W7.ld, ops._p(hot.hot_rows), 0, None, ops._p(H0.t), ops._p(Hd.t), H0.ld, 30,
                                             None, 1, 0.5, None, ops._p(mask), 1, 0, None, 0, 0, ops._stream())
    assert rc == -3                                        # GEOGCN_E_ALIGN


def test_gemm_asymmetric_detects_transposes(dev):
    """A = I check with an asymmetric B (guide rule: symmetric inputs hide row/col swaps)."""
    from geographconv_amd import ops
    K = 96
    A = np.eye(K, dtype=np.float32)
    B = (np.arange(K * 80).reshape(K, 80) % 97).astype(np.float32)
    got = ops.gemm(ops.DMat.from_numpy(A, dev), ops.DMat.from_numpy(B, dev)).numpy()
    assert np.array_equal(got, B)
    got = ops.gemm(ops.DMat.from_numpy(A, dev), ops.DMat.from_numpy(B, dev), transA=True).numpy()
    assert np.array_equal(got, B)


def test_gemm_rejects_bad_arguments(dev):
    from geographconv_amd import _ffi, ops
    lib = _ffi.lib()
    A = ops.DMat.from_numpy(_rand((8, 8), 1), dev)
    rc = lib.geogcn_gemm_f32(1, 1, 8, 8, 8, ops._p(A.t), 8, ops._p(A.t), 8, ops._p(A.t), 8, None, 0, 0, 0, None, 0,
                             ops._stream())
    assert rc == -4
    rc = lib.geogcn_gemm_f32(0, 0, 8, 8, 8, ops._p(A.t), 6, ops._p(A.t), 8, ops._p(A.t), 8, None, 0, 0, 0, None, 0,
                             ops._stream())
    assert rc in (-2, -3)


def test_sigmoid_epilogue_accuracy(dev):
    """The gate nonlinearity (T.nnet.sigmoid, gcnmodel.py:286) is evaluated with v_exp_f32 / v_rcp_f32 on
    a range-reduced argument: <= 3 ulp of the fp64 result over the whole float range, exact limits."""
    from geographconv_amd import ops
    x = np.concatenate([np.linspace(-40, 40, 200001), _rand((100000,), 5) * 6, [0.0, -0.0, 87.0, -87.0, 88.8, -88.8,
                        103.0, -103.9, 150.0, -150.0, 1e4, -1e4, 3e38, -3e38, np.inf, -np.inf]]).astype(np.float32)
    n = x.size // 4 * 4
    x = x[-n:].reshape(-1, 4)
    got = ops.bias_act(ops.DMat.from_numpy(x, dev), None, ops.ACT_SIGMOID).numpy().astype(np.float64)
    with np.errstate(over='ignore'):
        ref = 1.0 / (1.0 + np.exp(-x.astype(np.float64)))
    ulp = np.spacing(ref.astype(np.float32)).astype(np.float64)
    fin = ref > 1e-37                     # below: fp32 denormals, flushed (the reference's exp underflows too)
    assert np.all(np.abs(got - ref)[fin] <= 3 * ulp[fin])
    assert np.all(got[~fin] <= 2e-37) and np.all(got >= 0) and np.all(got <= 1)
    assert got[x == np.inf].min() == 1.0 and got[x == -np.inf].max() == 0.0
    nan = ops.bias_act(ops.DMat.from_numpy(np.full((1, 4), np.nan, np.float32), dev), None, ops.ACT_SIGMOID).numpy()
    assert np.all(np.isnan(nan))


def test_tanh_accuracy(dev):
    """The library's tanh (csrc/common.h apply_act<GEOGCN_ACT_TANH>: libm's tanhf -- a 22-instruction replacement, polynomial below
    |x| = 0.6 and 1 - 2 / (exp(2|x|) + 1) above, passed this test and the whole parity suite in round 6 and changed no kernel's time:
    profiles/r06_fast_tanh_ab.txt; not kept; every epilogue and elementwise kernel calls it) against float64 over every binade from 2^-126 to 2^4, both signs, the branch boundary, the saturation
    point and the special values: at most 3 units in the last place of the float32 result; odd; monotone on a fine grid; -0, +-inf, NaN."""
    from geographconv_amd import ops
    rng = np.random.RandomState(0)
    parts = [np.ldexp(1.0 + rng.rand(4096), e) for e in range(-126, 5)]                       # every binade
    parts += [np.linspace(0.55, 0.65, 200001), np.linspace(0.0, 12.0, 1200001), np.array([0.6, np.nextafter(np.float32(0.6), 0), 9.99, 10.0, 10.01, 88.0, 1e30])]
    x = np.concatenate(parts).astype(np.float32)
    x = np.concatenate([x, -x])
    F = 1024
    n = -(-x.size // F)
    X = np.zeros((n, F), np.float32)
    X.reshape(-1)[:x.size] = x
    zero = torch.zeros(F, device=dev)
    got = ops.bias_act(ops.DMat.from_numpy(X, dev), zero, ops.ACT_TANH).numpy().reshape(-1)[:x.size]
    ref = np.tanh(x.astype(np.float64))
    ulp = np.abs(got.astype(np.float64) - ref) / np.spacing(np.abs(ref).astype(np.float32)).astype(np.float64)
    assert ulp.max() <= 3.0, (float(ulp.max()), float(x[ulp.argmax()]))
    assert np.array_equal(got[:x.size // 2], -got[x.size // 2:])                             # odd
    grid = np.linspace(-12, 12, 2000001).astype(np.float32)
    G = np.zeros((-(-grid.size // F), F), np.float32)
    G.reshape(-1)[:grid.size] = grid
    g = ops.bias_act(ops.DMat.from_numpy(G, dev), zero, ops.ACT_TANH).numpy().reshape(-1)[:grid.size]
    assert np.all(np.diff(g) >= -2.4e-7) and np.all(np.abs(g) <= 1.0)                        # monotone to 2 ulp of 1, never beyond 1
    sp = np.zeros((1, F), np.float32)
    sp[0, :5] = [-0.0, np.inf, -np.inf, np.nan, 0.0]
    s = ops.bias_act(ops.DMat.from_numpy(sp, dev), zero, ops.ACT_TANH).numpy()[0]
    assert s[0] == 0 and s[1] == 1 and s[2] == -1 and np.isnan(s[3]) and s[4] == 0          # (-0 + the zero bias is +0 before the activation sees it)


@pytest.mark.parametrize("n,F", [(1000, 300), (257, 129), (3, 5)])
def test_highway_and_tanh_kernels(dev, n, F):
    from geographconv_amd import ops
    T = O.sigmoid(_rand((n, F), 1)).astype(np.float32)
    Hc = np.tanh(_rand((n, F), 2))
    H = _rand((n, F), 3)
    G = _rand((n, F), 4)
    d = lambda a: ops.DMat.from_numpy(a, dev)
    out = ops.highway_fwd(d(T), d(Hc), d(H)).numpy()
    ref = T * Hc + (np.float32(1) - T) * H
    assert np.allclose(out, ref, rtol=1e-6, atol=1e-7)
    dS, dU, dHc = ops.highway_bwd(d(G), d(T), d(Hc), d(H))
    assert np.allclose(dS.numpy(), G * T * (1 - Hc * Hc), rtol=1e-5, atol=1e-7)
    assert np.allclose(dU.numpy(), G * (Hc - H) * T * (1 - T), rtol=1e-5, atol=1e-7)
    assert np.allclose(dHc.numpy(), G * (1 - T), rtol=1e-6, atol=1e-7)
    assert torch.all(dS.t[:, F:ops.pad4(F)] == 0)
    # fused bias gradients: same dS / dU, plus deterministic column sums
    bS = torch.zeros(ops.pad4(F), device=dev)
    bU = torch.zeros(ops.pad4(F), device=dev)
    dS2, dU2, dHc2 = ops.highway_bwd(d(G), d(T), d(Hc), d(H), dbS=bS, dbU=bU)
    assert np.array_equal(dS2.numpy(), dS.numpy()) and np.array_equal(dU2.numpy(), dU.numpy())
    assert np.array_equal(dHc2.numpy(), dHc.numpy())
    refS = dS.numpy().astype(np.float64).sum(axis=0)
    refU = dU.numpy().astype(np.float64).sum(axis=0)
    assert np.all(np.abs(bS.cpu().numpy()[:F] - refS) <= 1e-6 * np.abs(dS.numpy()).sum(axis=0) + 1e-6)
    assert np.all(np.abs(bU.cpu().numpy()[:F] - refU) <= 1e-6 * np.abs(dU.numpy()).sum(axis=0) + 1e-6)
    bS2 = torch.zeros_like(bS)
    ops.highway_bwd(d(G), d(T), d(Hc), d(H), dbS=bS2, dbU=torch.zeros_like(bU))
    assert torch.equal(bS, bS2)
    # dS stored as bf16 (bf16 configuration): the bits of cast_bf16 over the fp32 dS, whole pitch written, everything else as is
    for kw in (dict(), dict(dbS=torch.zeros_like(bS), dbU=torch.zeros_like(bU))):
        h = ops.HMat(n, F, dev)
        h.t.fill_(7.0)
        dSh, dUh, dHh = ops.highway_bwd(d(G), d(T), d(Hc), d(H), dS=h, dS_bf16=True, **kw)
        assert dSh is h and torch.equal(h.t, ops.cast_bf16(dS).t)
        assert torch.equal(dUh.t, dU.t) and torch.equal(dHh.t, dHc.t)
        if kw:
            assert torch.equal(kw['dbS'], bS) and torch.equal(kw['dbU'], bU)
    # tanh backward, with and without the dropout mask folded in
    got = ops.tanh_bwd(d(G), d(Hc)).numpy()
    assert np.allclose(got, G * (1 - Hc * Hc), rtol=1e-5, atol=1e-7)
    mask = (np.random.RandomState(5).rand(n, F) < 0.5).astype(np.uint8)
    tm = torch.from_numpy(mask).to(dev)
    got = ops.tanh_bwd(d(G), d(Hc), keep_mask=tm, scale=2.0).numpy()
    assert np.allclose(got, G * mask * 2.0 * (1 - Hc * Hc), rtol=1e-5, atol=1e-7)
    # bias + act
    b = _rand((F,), 6)
    tb = torch.from_numpy(np.pad(b, (0, ops.pad4(F) - F))).to(dev)
    got = ops.bias_act(d(H), tb, ops.ACT_TANH)
    assert np.allclose(got.numpy(), np.tanh(H + b), rtol=1e-5, atol=1e-6)
    assert torch.all(got.t[:, F:] == 0)
    # dropout apply == x * mask / (1-p)
    got = ops.dropout_apply(d(H), tm, 0.5).numpy()
    assert np.array_equal(got, H * mask * np.float32(2.0))


@pytest.mark.parametrize("n,F,act", [(5000, 300, 1), (257, 129, 2), (3, 5, 0), (777, 300, 4)])
def test_act_bwd_colsum_fused(dev, n, F, act):
    """geogcn_act_bwd_colsum_f32 = act_bwd followed by colsum (the fused kernel for tanh / sigmoid / none at the
    natural pitch, the two-pass fallback otherwise), with and without the dropout mask; deterministic."""
    from geographconv_amd import ops
    G = ops.DMat.from_numpy(_rand((n, F), 1), dev)
    Y = ops.DMat.from_numpy(np.tanh(_rand((n, F), 2)), dev)
    mask = torch.from_numpy((np.random.RandomState(3).rand(n, F) < 0.5).astype(np.uint8)).to(dev)
    for km, sc in ((None, 1.0), (mask, 2.0)):
        ref = ops.act_bwd(G, Y, act, keep_mask=km, scale=sc)
        db_ref = ops.colsum(ref)
        db = torch.zeros(ops.pad4(F), dtype=torch.float32, device=dev)
        got = ops.act_bwd_colsum(G, Y, act, db, keep_mask=km, scale=sc)
        assert torch.equal(got.t, ref.t)
        col = np.abs(ref.numpy()).sum(0)
        assert np.all(np.abs(db[:F].cpu().numpy() - db_ref[:F].cpu().numpy()) <= 2e-6 * col + 1e-7)
        db2 = torch.zeros_like(db)
        ops.act_bwd_colsum(G, Y, act, db2, keep_mask=km, scale=sc)
        assert torch.equal(db, db2)
        # line-aligned output pitch (what the graph convolutions ask for)
        out = ops.DMat.empty(n, F, dev, ld=ops.gather_ld(F))
        ops.act_bwd_colsum(G, Y, act, db2, out=out, keep_mask=km, scale=sc)
        assert torch.equal(out.t[:, :ops.pad4(F)], ref.t) and torch.equal(db, db2)


def test_colsum_gather_deterministic(dev):
    from geographconv_amd import ops
    X = _rand((70001, 300), 1)
    dX = ops.DMat.from_numpy(X, dev)
    s1 = ops.colsum(dX).cpu().numpy()[:300]
    s2 = ops.colsum(dX).cpu().numpy()[:300]
    assert np.array_equal(s1, s2)
    ref = X.astype(np.float64).sum(axis=0)
    assert np.all(np.abs(s1 - ref) <= 1e-6 * np.abs(X).sum(axis=0) + 1e-5)
    idx = torch.from_numpy(np.random.RandomState(2).randint(0, 70001, 999).astype(np.int32)).to(dev)
    got = ops.gather_rows(dX, idx).cpu().numpy()
    assert np.array_equal(got, X[idx.cpu().numpy()])


def test_philox_mask_statistics_and_reproducibility(dev):
    from geographconv_amd import ops
    m1 = ops.dropout_mask(4000, 300, 0.5, seed=77, offset=0, device=dev).cpu().numpy()
    m2 = ops.dropout_mask(4000, 300, 0.5, seed=77, offset=0, device=dev).cpu().numpy()
    m3 = ops.dropout_mask(4000, 300, 0.5, seed=78, offset=0, device=dev).cpu().numpy()
    assert np.array_equal(m1, m2) and not np.array_equal(m1, m3)
    assert set(np.unique(m1)) <= {0, 1}
    assert abs(m1.mean() - 0.5) < 5e-3
    m4 = ops.dropout_mask(4000, 300, 0.2, seed=1, offset=0, device=dev).cpu().numpy()
    assert abs(m4.mean() - 0.8) < 5e-3
    # no visible row/column structure
    assert abs(np.corrcoef(m1[:, 0], m1[:, 1])[0, 1]) < 0.06


@pytest.mark.parametrize("n,Cc", [(3000, 129), (1000, 256), (50, 930), (7, 2), (40, 1500)])
def test_softmax_ce_head(dev, n, Cc):
    from geographconv_amd import ops
    L = _rand((n, Cc), 1, scale=3.0)
    L[0, :] = 0.0                                   # all-ties row: argmax must be the FIRST index
    L[1, 5 % Cc] = L[1].max() + 1
    dL = ops.DMat.from_numpy(L, dev)
    am = torch.zeros(n, dtype=torch.int32, device=dev)
    P = ops.softmax_rows(dL, argmax=am)
    ref = O.softmax_rows(L)
    assert np.allclose(P.numpy(), ref, rtol=2e-6, atol=1e-9)
    assert np.array_equal(am.cpu().numpy(), L.argmax(-1))
    assert am[0].item() == 0
    assert torch.all(P.t[:, Cc:] == 0)
    rng = np.random.RandomState(3)
    idx = rng.choice(n, size=max(1, n // 2), replace=False).astype(np.int32)
    y = rng.randint(0, Cc, len(idx)).astype(np.int32)
    y[: len(y) // 3] = ref[idx[: len(y) // 3]].argmax(-1)        # make some hits
    ti, ty = torch.from_numpy(idx).to(dev), torch.from_numpy(y).to(dev)
    for amx in (am, None):
        out2 = ops.ce_metrics(P, ti, ty, argmax=amx).cpu().numpy()
        loss_ref, acc_ref, _ = O.metrics(ref.astype(np.float64), idx, y)
        assert abs(out2[0] / len(idx) - loss_ref) <= 2e-6 * abs(loss_ref) + 1e-6
        assert out2[1] == round(acc_ref * len(idx))
    D = ops.softmax_ce_bwd(P, ti, ty)
    refD = np.zeros_like(ref)
    refD[idx] = ref[idx]
    refD[idx, y] -= 1
    refD /= len(idx)
    assert np.allclose(D.numpy(), refD, rtol=1e-5, atol=1e-9)
    untouched = np.setdiff1d(np.arange(n), idx)
    assert np.all(D.numpy()[untouched] == 0)


@pytest.mark.parametrize("n,Cc", [(3000, 256), (500, 129), (40, 930)])
def test_softmax_ce_bwd_with_bias_gradient(dev, n, Cc):
    """geogcn_softmax_ce_bwd_db_f32: the same dlogits as geogcn_softmax_ce_bwd_f32 plus their column sums (the
    output layer's bias gradient) without a pass over the N x C matrix; deterministic."""
    from geographconv_amd import ops
    rng = np.random.RandomState(Cc)
    P = ops.softmax_rows(ops.DMat.from_numpy(_rand((n, Cc), 1), dev))
    idx = torch.from_numpy(rng.permutation(n)[:n * 2 // 3].astype(np.int32)).to(dev)
    y = torch.from_numpy(rng.randint(0, Cc, idx.numel()).astype(np.int32)).to(dev)
    ref = ops.softmax_ce_bwd(P, idx, y, inv_n=1.0 / idx.numel())
    db_ref = ref.numpy().astype(np.float64).sum(0)
    db = torch.full((ops.pad4(Cc),), 7.0, dtype=torch.float32, device=dev)
    got = ops.softmax_ce_bwd(P, idx, y, inv_n=1.0 / idx.numel(), db=db)
    assert torch.equal(got.t, ref.t)
    assert np.all(np.abs(db[:Cc].cpu().numpy() - db_ref) <= 2e-6 * np.abs(ref.numpy()).sum(0) + 1e-8)
    db2 = torch.zeros_like(db)
    ops.softmax_ce_bwd(P, idx, y, inv_n=1.0 / idx.numel(), db=db2)
    assert torch.equal(db[:Cc], db2[:Cc])
    empty = torch.zeros(0, dtype=torch.int32, device=dev)
    ops.softmax_ce_bwd(P, empty, empty, inv_n=1.0, db=db2)
    assert torch.all(db2[:Cc] == 0)


def test_adam_matches_lasagne_formula(dev):
    from geographconv_amd import ops
    n = 100003
    p = _rand((n,), 1)
    regmask = (np.arange(n) % 3 != 0).astype(np.float32)
    st = O.AdamState([p])
    tp = torch.from_numpy(p.copy()).to(dev)
    tm, tv = torch.zeros_like(tp), torch.zeros_like(tp)
    tr = torch.from_numpy(regmask).to(dev)
    cur = [p.copy()]
    for t in range(1, 4):
        g = _rand((n,), 10 + t, scale=0.1)
        reg = 1e-3
        g_ref = g + regmask * np.float32(reg) * (np.sign(cur[0]) + np.float32(2) * cur[0])
        cur = O.adam_step(cur, [g_ref.astype(np.float32)], st)
        ops.adam_step(tp, torch.from_numpy(g).to(dev), tm, tv, tr, 2e-3, 0.9, 0.999, 1e-8, t, l1=reg, l2=reg)
        assert np.allclose(tp.cpu().numpy(), cur[0], rtol=1e-5, atol=1e-7)
    pen = ops.reg_penalty(tp, tr, 1e-3, 1e-3).item()
    pc = tp.cpu().numpy().astype(np.float64)
    ref = 1e-3 * (np.abs(pc) * regmask).sum() + 1e-3 * (pc * pc * regmask).sum()
    assert abs(pen - ref) <= 1e-5 * ref


def test_cmu_shape_spmm_full_size(dev):
    """Config 2 operand sizes (CMU shape), all three SpMM flavours of the path."""
    from geographconv_amd import ops
    s = synth.CMU
    A = synth.powerlaw_ahat(s.N, s.E_target)
    X = synth.bow_x(s.N, s.V, s.mean_nnz)
    H = np.random.RandomState(1).randn(s.N, 300).astype(np.float32)
    W0 = _rand((s.V, 300), 2, scale=0.05)
    dA, dX = ops.CSR(A, dev), ops.CSR(X, dev)
    dXt = ops.CSR(sps.csr_matrix(X.T), dev)
    assert dA.n_long_rows > 0
    got = ops.spmm(dA, ops.DMat.from_numpy(H, dev)).numpy()
    assert np.allclose(got, O.spmm(A, H), rtol=1e-5, atol=2e-6)
    got = ops.spmm(dX, ops.DMat.from_numpy(W0, dev)).numpy()
    assert np.allclose(got, O.spmm(X, W0), rtol=1e-5, atol=2e-6)
    got = ops.spmm(dXt, ops.DMat.from_numpy(H, dev)).numpy()
    ref = O.spmm_t(X, H)
    assert np.allclose(got, ref, rtol=1e-4, atol=2e-5)
    # the hybrid transpose product the model uses: dense head panel (MFMA) + sparse tail (gather)
    sx = ops.SparseOperand.from_scipy(X, dev)
    assert sx.head_dense is not None and 16 <= sx.head_dense.F <= tuning.DENSE_HEAD_MAX_COLS
    assert sx.bwd.nnz + int((X[:, sx.head_idx.cpu().numpy()]).nnz) == X.nnz
    ref64 = (X.T.astype(np.float64) @ H.astype(np.float64))
    got = ops.spmm_t(sx, ops.DMat.from_numpy(H, dev)).numpy()
    mag = np.asarray(abs(X).T @ np.abs(H))
    assert np.all(np.abs(got - ref64) <= 3e-6 * mag + 1e-5), np.abs(got - ref64).max()
    got2 = ops.spmm_t(sx, ops.DMat.from_numpy(H, dev)).numpy()
    assert np.array_equal(got, got2)                     # deterministic


def test_randomised_shape_sweep(dev):
    """Seeded sweep over awkward shapes (every K4 / lane-group / tile / tail path of the SpMM and GEMM kernels):
    fp32 against fp64 references with the accumulation-error envelope, bf16 modes against their own envelopes."""
    from geographconv_amd import ops
    rng = np.random.RandomState(20260928)
    # ---- SpMM (fp32 and bf16 operand, bias + tanh epilogue on half of the cases)
    for case in range(36):
        n_rows, n_cols = int(rng.randint(1, 400)), int(rng.randint(1, 400))
        F = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 300, 320, 321, 511,
                            600, 640, 641, 900, 1023, 1024]))
        dens = float(rng.choice([0.02, 0.2, 0.9]))
        A = sps.random(n_rows, n_cols, density=dens, random_state=rng, format='csr', dtype=np.float32)
        if n_rows > 3:                                          # one long row (chunked) and one empty row
            A = A.tolil()
            A[0, :] = rng.randn(n_cols).astype(np.float32)
            A[n_rows - 1, :] = 0
            A = sps.csr_matrix(A, dtype=np.float32)
        A.eliminate_zeros()
        A.sort_indices()
        B = rng.randn(n_cols, F).astype(np.float32)
        use_epi = case % 2 == 0
        bias = rng.randn(F).astype(np.float32) if use_epi else None
        dA = ops.CSR(A, dev)
        dB = ops.DMat.from_numpy(B, dev)
        db = torch.from_numpy(np.pad(bias, (0, ops.pad4(F) - F))).to(dev) if use_epi else None
        for bf in (False, True):
            Bop = ops.cast_bf16(dB) if bf else dB
            Bref = (torch.from_numpy(B).to(torch.bfloat16).float().numpy() if bf else B).astype(np.float64)
            ref = np.asarray(A.astype(np.float64) @ Bref)
            mag = np.asarray(abs(A).astype(np.float64) @ np.abs(Bref))
            if use_epi:
                ref, mag = np.tanh(ref + bias), mag + np.abs(bias)
            out = ops.spmm(dA, Bop, bias=db, act=ops.ACT_TANH if use_epi else ops.ACT_NONE)
            err = np.abs(out.numpy() - ref)
            assert np.all(err <= 2e-6 * mag + 2e-6), (case, n_rows, n_cols, F, bf, err.max())
            assert torch.all(out.t[:, F:] == 0)
    # ---- GEMM
    for case in range(36):
        M, N, K = int(rng.randint(1, 1500)), int(rng.randint(1, 700)), int(rng.randint(1, 700))
        if case % 6 == 0:
            M, N, K = int(rng.choice([1, 127, 128, 129, 4096])), int(rng.choice([1, 159, 160, 161, 320, 321])), int(rng.choice([1, 31, 32, 33, 300]))
        A = rng.randn(M, K).astype(np.float32)
        B = rng.randn(K, N).astype(np.float32)
        ref = A.astype(np.float64) @ B.astype(np.float64)
        mag = np.abs(A).astype(np.float64) @ np.abs(B)
        dA, dB = ops.DMat.from_numpy(A, dev), ops.DMat.from_numpy(B, dev)
        dBt = ops.DMat.from_numpy(np.ascontiguousarray(B.T), dev)
        dAt = ops.DMat.from_numpy(np.ascontiguousarray(A.T), dev)
        for prec, tol in (('f32', 2e-6), ('bf16x3', 2e-6), ('bf16', 2 ** -7)):
            for name, got in (('nn', ops.gemm(dA, dB, precision=prec)), ('nt', ops.gemm(dA, dBt, transB=True, precision=prec)),
                              ('tn', ops.gemm(dAt, dB, transA=True, precision=prec))):
                err = np.abs(got.numpy() - ref)
                assert np.all(err <= tol * mag + 2e-6), (case, M, N, K, prec, name, err.max())
                assert torch.all(got.t[:, N:] == 0)
