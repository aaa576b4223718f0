"""NumPy TEST DOUBLE of geographconv_amd.ops -- lives under tests/ on purpose.

It exists so that the multi-rank *communication* logic (row partition, in-place all-gather slots,
gradient / metric all-reduce, index splitting) can run on CPU under gloo with world_size 2.  It is
NOT a product backend: nothing in geographconv_amd/ imports it, and the arithmetic here is the
oracle's (numpy / scipy), i.e. the checker, applied to CPU tensors."""
import numpy as np
import scipy.sparse as sps
import torch

from geographconv_amd.ops import DMat, HMat, Panels, gather_ld, pad4  # noqa: F401  (pure containers, device-agnostic)

ACT_NONE, ACT_TANH, ACT_SIGMOID = 0, 1, 2


def require_gpu():
    return None


def _act(x, act):
    if act == ACT_TANH:
        return np.tanh(x)
    if act == ACT_SIGMOID:
        return (1.0 / (1.0 + np.exp(-x))).astype(np.float32)
    if act == 3:
        return (1.0507009873554805 * np.where(x > 0, x, 1.6732632423543772 * (np.exp(x) - 1))).astype(np.float32)
    if act == 4:
        return np.maximum(x, 0)
    return x


def _v(m: DMat):
    return m.t.numpy()[:, :m.F]


class CSR:
    def __init__(self, m, device=None, long_row_nnz=256, chunk_nnz=128, **kw):
        self.m = sps.csr_matrix(m).astype(np.float32)
        self.shape = self.m.shape
        self.nnz = self.m.nnz
        self.rowptr_host = self.m.indptr.astype(np.int32)


class SparseOperand:
    def __init__(self, fwd, bwd, symmetric):
        self.fwd, self.bwd, self.symmetric, self.shape = fwd, bwd, symmetric, fwd.shape

    @staticmethod
    def from_scipy(m, device, need_transpose=True, **kw):
        m = sps.csr_matrix(m).astype(np.float32)
        return SparseOperand(CSR(m), CSR(sps.csr_matrix(m.T)), False)


def bf16_gather(precision=None):
    return False          # the double computes everything in fp32


def highway_bwd_bf16_ok(G, with_bias):
    return False


def spmm(A, B, out=None, bias=None, act=ACT_NONE, F=None, out_col0=0):
    F = B.F if F is None else F
    if out is None:
        out = DMat(A.shape[0], F, B.device)
    r = np.asarray(A.m @ B.t.float().numpy()[:A.shape[1], :F])
    if bias is not None:
        r = r + bias.numpy()[:F]
    out.t.numpy()[:, out_col0:out_col0 + F] = _act(r.astype(np.float32), act)        # (F may be narrower than the output buffer)
    return out


def spmm_t(x, G, out=None, precision=None):
    return spmm(x.bwd, G, out=out)


def spmm_x(x, W, out=None, bias=None, act=ACT_NONE):
    return spmm(x.fwd, W, out=out, bias=bias, act=act)


def spmm_x_dropout(x, W, bias, act, p, mask_in=None, seed=0, offset=0, calls_dev=None, per_call=0, base=0):
    return None          # (the fused product + dropout launch is a property of the HIP backend: callers fall back)


def gemm(A, B, out=None, transA=False, transB=False, bias=None, act=ACT_NONE, accumulate=False, precision=None, gate_carry=None):
    if gate_carry is not None:
        assert not (transA or accumulate or bias is not None or act != ACT_NONE)
        c = gate_carry.dense()
        if out is None:
            out = c
        else:
            out.t.copy_(c.t)
        accumulate = True
    a = _v(A).T if transA else _v(A)
    b = _v(B).T if transB else _v(B)
    r = a @ b
    if bias is not None:
        r = r + bias.numpy()[:r.shape[1]]
    r = _act(r.astype(np.float32), act)
    if isinstance(out, Panels):                      # the all-to-all's send layout, written by the product itself
        o = out.t.numpy().reshape(out.W, out.R, out.wp)
        for q in range(out.W):
            c0, c1 = q * out.wp, min(r.shape[1], (q + 1) * out.wp)
            if c1 > c0:
                o[q, :r.shape[0], :c1 - c0] = r[:, c0:c1]
        return out
    if out is None:
        out = DMat(r.shape[0], r.shape[1], A.device)
    if accumulate:
        _v(out)[...] += r
    else:
        _v(out)[...] = r
    return out


GEMM_PRECISION = 'f32'


def gemm_dual(A, B0, B1, out0=None, out1=None, transA=False, bias0=None, act0=ACT_NONE, bias1=None, act1=ACT_NONE, precision=None):
    return (gemm(A, B0, out=out0, transA=transA, bias=bias0, act=act0),
            gemm(A, B1, out=out1, transA=transA, bias=bias1, act=act1))


class GateCarry:
    def __init__(self, G, T):
        self.G, self.T = G, T

    def dense(self):
        out = DMat(self.G.n, self.G.F, self.G.device)
        _v(out)[...] = _v(self.G) * (1 - _v(self.T))
        return out


def gemm_gated_native(n, F, precision=None):
    return True


def kcat_gated_native(n, F, precision=None):
    return True              # (the double takes the gated form at every size: the host logic is what it exercises)


def gemm_kcat(A0, B0, A1, B1, out=None, transB=False, accumulate=False, gate_carry=None, tanh_bwd=None, precision=None):
    if tanh_bwd is not None:
        Y, keep, scale = tanh_bwd
        tmp = gemm_kcat(A0, B0, A1, B1, transB=transB, gate_carry=gate_carry)
        _v(out)[...] = _v(tmp) * (keep.numpy().astype(np.float32) * np.float32(scale)) * (1 - _v(Y) * _v(Y))
        return out
    if gate_carry is not None:
        assert not accumulate
        c = gate_carry.dense()
        if out is None:
            out = c
        else:
            out.t.copy_(c.t)
        accumulate = True
    out = gemm(A0, B0, out=out, transB=transB, accumulate=accumulate)
    return gemm(A1, B1, out=out, transB=transB, accumulate=True)


def spmm_softmax_ok(B, F):
    return False          # (the double keeps the two passes)


def spmm_highway(A, B, bias, T, H, Hc=None, Hout=None):
    Hc = spmm(A, B, out=Hc, bias=bias, act=ACT_TANH, F=H.F)
    return Hc, highway_fwd(T, Hc, H, out=Hout)


def act_bwd_colsum(G, Y, act, db, out=None, keep_mask=None, scale=1.0):
    out = act_bwd(G, Y, act, out=out, keep_mask=keep_mask, scale=scale)
    db.numpy()[:G.F] = _v(out).sum(axis=0)
    return out


def bias_act(X, bias, act, out=None):
    out = DMat(X.n, X.F, X.device) if out is None else out
    r = _v(X) + (0 if bias is None else bias.numpy()[:X.F])
    _v(out)[...] = _act(r.astype(np.float32), act)
    return out


def highway_fwd(T, Hc, H, out=None):
    out = DMat(H.n, H.F, H.device) if out is None else out
    _v(out)[...] = _v(T) * _v(Hc) + (np.float32(1) - _v(T)) * _v(H)
    return out


def highway_bwd(G, T, Hc, H, dS=None, dU=None, dHcarry=None, dbS=None, dbU=None, carry=True):
    mk = lambda: DMat(G.n, G.F, G.device)
    dS, dU = dS or mk(), dU or mk()
    g, t, hc, h = _v(G), _v(T), _v(Hc), _v(H)
    _v(dS)[...] = g * t * (1 - hc * hc)
    _v(dU)[...] = g * (hc - h) * t * (1 - t)
    if carry:
        dHcarry = dHcarry or mk()
        _v(dHcarry)[...] = g * (1 - t)
    else:
        dHcarry = None
    if dbS is not None:
        dbS.numpy()[:G.F] = _v(dS).sum(axis=0)
        dbU.numpy()[:G.F] = _v(dU).sum(axis=0)
    return dS, dU, dHcarry


def act_bwd(G, Y, act, out=None, keep_mask=None, scale=1.0):
    out = DMat(G.n, G.F, G.device) if out is None else out
    g, y = _v(G), _v(Y)
    if keep_mask is not None:
        g = g * keep_mask.numpy().astype(np.float32) * np.float32(scale)
    d = (1 - y * y) if act == ACT_TANH else (y * (1 - y) if act == ACT_SIGMOID else 1.0)
    _v(out)[...] = g * d
    return out


def add_inplace(X, Y):
    Y.t += X.t
    return Y


def colsum_rowblocks(X, out):
    return colsum(X, out=out)


def colsum(X, out=None):
    if out is None:
        out = torch.zeros(pad4(X.F), dtype=torch.float32)
    out.numpy()[:X.F] = _v(X).sum(axis=0)
    return out


def dropout_mask(n, F, p, seed, offset, device, out=None):
    # deterministic per (seed, offset); NOT the Philox stream of the HIP kernel
    rng = np.random.RandomState((int(seed) + 7919 * int(offset)) % (2 ** 31))
    return torch.from_numpy((rng.rand(n, F) < (1 - p)).astype(np.uint8))


def dropout_apply(X, keep_mask, p, out=None):
    out = DMat(X.n, X.F, X.device) if out is None else out
    _v(out)[...] = _v(X) * keep_mask.numpy().astype(np.float32) * np.float32(1.0 / (1.0 - p))
    return out


def softmax_rows(L, out=None, argmax=None):
    out = DMat(L.n, L.F, L.device) if out is None else out
    x = _v(L)
    e = np.exp(x - x.max(axis=1, keepdims=True))
    _v(out)[...] = e / e.sum(axis=1, keepdims=True)
    if argmax is not None:
        argmax.numpy()[...] = x.argmax(-1).astype(np.int32)
    return out


def ce_metrics(P, idx, y, argmax=None, out2=None):
    if out2 is None:
        out2 = torch.zeros(2, dtype=torch.float32)
    rows = _v(P)[idx.numpy()]
    yy = y.numpy()
    out2.numpy()[0] = (-np.log(rows[np.arange(len(yy)), yy])).sum() if len(yy) else 0.0
    out2.numpy()[1] = float((rows.argmax(-1) == yy).sum()) if len(yy) else 0.0
    return out2


def softmax_ce_bwd(P, idx, y, out=None, inv_n=None, db=None):
    out = DMat(P.n, P.F, P.device) if out is None else out
    out.t.zero_()
    i, yy = idx.numpy(), y.numpy()
    inv_n = 1.0 / max(1, len(i)) if inv_n is None else inv_n
    d = _v(out)
    d[i] = _v(P)[i]
    d[i, yy] -= 1
    d[i] *= np.float32(inv_n)
    if db is not None:
        db.numpy()[:P.F] = d.sum(axis=0)
    return out


def softmax_ce_rows_bwd(P, idx, y, inv_n, db, out=None):
    i = idx.cpu().numpy().astype(np.int64)
    g = _v(P)[i].copy()
    g[np.arange(len(i)), y.cpu().numpy().astype(np.int64)] -= 1.0
    g *= np.float32(inv_n)
    out = DMat.from_numpy(g.astype(np.float32), P.device) if out is None else out
    db[:P.F] = torch.from_numpy(g.sum(axis=0).astype(np.float32))
    return out


def gather_rows(X, idx, out=None):
    return torch.from_numpy(_v(X)[idx.numpy()].copy())


def pack_rows(X, idx, out):
    if idx.numel():
        out.copy_(X.t[idx.long()])
    return out


def pack_panels(X, R, W, wp, out):
    o = out.numpy().reshape(W, R, wp)
    o[...] = 0
    x = _v(X)
    for q in range(W):
        c0, c1 = q * wp, min(X.F, (q + 1) * wp)
        if c1 > c0:
            o[q, :X.n, :c1 - c0] = x[:, c0:c1]
    return out


def unpack_panels(inp, R, W, wp, out):
    i = inp.numpy().reshape(W, R, wp)
    y = _v(out)
    for q in range(W):
        c0, c1 = q * wp, min(out.F, (q + 1) * wp)
        if c1 > c0:
            y[:, c0:c1] = i[q, :out.n, :c1 - c0]
    return out


def adam_step(p, g, m, v, regmask, lr, b1, b2, eps, t, l1=0.0, l2=0.0):
    P, G, M, V = p.numpy(), g.numpy(), m.numpy(), v.numpy()
    if l1 or l2:
        G += regmask.numpy() * (np.float32(l1) * np.sign(P) + np.float32(2 * l2) * P)
    a_t = np.float32(lr) * np.sqrt(np.float32(1) - np.float32(b2) ** np.float32(t)) / (np.float32(1) - np.float32(b1) ** np.float32(t))
    M[...] = np.float32(b1) * M + np.float32(1 - b1) * G
    V[...] = np.float32(b2) * V + np.float32(1 - b2) * G * G
    P[...] = P - a_t * M / (np.sqrt(V) + np.float32(eps))


def reg_penalty(p, regmask, l1, l2, out=None):
    if out is None:
        out = torch.zeros(1, dtype=torch.float32)
    P = p.numpy().astype(np.float64)
    out.numpy()[0] = (regmask.numpy() * (l1 * np.abs(P) + l2 * P * P)).sum()
    return out
