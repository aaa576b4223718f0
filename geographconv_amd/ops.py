"""Device containers + thin call wrappers over the C ABI (include/geogcn.h).

torch is used for three things only: device memory (tensors), the current HIP stream, and
``torch.distributed``.  Every arithmetic op below is a call into libgeogcn.so; if the library or
a GPU is missing the call raises (no eager / CPU fallback)."""
from __future__ import annotations

import ctypes as C

import numpy as np
import scipy.sparse as sps
import torch

from . import _ffi, tuning
from ._ffi import ACT_NONE, ACT_RELU, ACT_SELU, ACT_SIGMOID, ACT_TANH, check  # noqa: F401


class StepTimers:
    """Measurement aid (bench.py): while an instance is installed with `with StepTimers() as t:`, the wrappers named in
    `_timed` record a torch.cuda.Event pair around their launches on the current stream -- the kernels timed INSIDE a real
    training step, on the step's own operands and cache state, instead of isolated re-runs.  Off (None) in normal use: the
    wrappers then pay one attribute test."""
    active = None

    def __init__(self):
        self.events = {}

    def __enter__(self):
        StepTimers.active = self
        return self

    def __exit__(self, *exc):
        StepTimers.active = None

    def ms(self):
        """-> {label: [milliseconds per call]} (synchronises)."""
        torch.cuda.synchronize()
        return {k: [a.elapsed_time(b) for a, b in v] for k, v in self.events.items()}


def _timed(label):
    def deco(fn):
        def wrapper(*args, **kw):
            t = StepTimers.active
            if t is None:
                return fn(*args, **kw)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            out = fn(*args, **kw)
            if out is None:                # (a fused path that declined: nothing ran)
                return out
            b.record()
            name = label(*args, **kw) if callable(label) else label
            t.events.setdefault(name, []).append((a, b))
            return out
        wrapper.__name__, wrapper.__doc__ = fn.__name__, fn.__doc__
        return wrapper
    return deco


def pad4(F: int) -> int:
    return (int(F) + 3) // 4 * 4


def gather_ld(F: int) -> int:
    """Pitch for a matrix whose ROWS an SpMM gathers: whole 128-byte lines per row (300 -> 320 floats),
    so a gathered row never shares a cache line with its neighbours -- the precondition for streaming
    the non-hub rows with non-temporal loads without re-fetching shared boundary lines."""
    F = int(F)
    return (F + 31) // 32 * 32 if F >= 64 else pad4(F)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def require_gpu():
    if not torch.cuda.is_available():
        raise _ffi.GeoGcnError("the GCN hot path runs on an MI355X only: no GPU visible "
                               "(torch.cuda.is_available() is False) and there is no CPU fallback")


class DMat:
    """Row-major fp32 device matrix with a 4-float-aligned pitch; pad columns are kept zero
    (geogcn.h convention) so a padded matrix is a valid reduction operand."""
    __slots__ = ('t', 'n', 'F')

    def __init__(self, n, F, device=None, t=None, ld=None):
        self.n, self.F = int(n), int(F)
        if t is None:
            t = torch.zeros((self.n, int(ld) if ld else pad4(F)), dtype=torch.float32, device=device)
        assert t.dtype == torch.float32 and t.is_contiguous() and t.dim() == 2 and t.shape[0] == self.n
        assert t.shape[1] >= pad4(F) and t.shape[1] % 4 == 0
        self.t = t

    @property
    def ld(self):
        return self.t.shape[1]

    @property
    def device(self):
        return self.t.device

    @staticmethod
    def from_numpy(a, device):
        a = np.ascontiguousarray(a, dtype=np.float32)
        if a.ndim == 1:
            a = a[None, :]
        m = DMat(a.shape[0], a.shape[1], device)
        m.t[:, :a.shape[1]].copy_(torch.from_numpy(a))
        return m

    def numpy(self):
        a = self.t[:, :self.F].cpu().numpy()
        return a.copy() if self.t.device.type == 'cpu' else a      # never alias a reusable buffer

    def like(self):
        return DMat.empty(self.n, self.F, self.t.device)

    @staticmethod
    def empty(n, F, device, ld=None):
        """Uninitialised when there are no pad columns (the producer writes every element);
        zero-filled otherwise so the pads honour the zero convention.  A matrix with the line-aligned
        gather pitch (`ld=gather_ld(F)`) is only ever read through the SpMM / GEMM, which never touch
        columns >= roundup4(F), so its extra pad columns may stay uninitialised."""
        ld = int(ld) if ld else pad4(F)
        if pad4(F) == int(F):
            return DMat(n, F, t=torch.empty((int(n), ld), dtype=torch.float32, device=device))
        return DMat(n, F, device, ld=ld)

    def rows(self, r0, r1):
        """View of a row range (shares storage)."""
        return DMat(r1 - r0, self.F, t=self.t[r0:r1])

    def copy_from(self, other):
        """Copy the logical columns of `other` (pitches may differ)."""
        if other.ld == self.ld:
            self.t.copy_(other.t)
        else:
            w = pad4(self.F)
            self.t[:, :w].copy_(other.t[:, :w])
        return self


class HMat:
    """Row-major bfloat16 device matrix (the gathered operand of the SpMM in the bf16 configuration,
    include/geogcn.h geogcn_spmm_csr_bf16b).  Pitch = gather_ld(F): 128-byte multiples, pads zero."""
    __slots__ = ('t', 'n', 'F')

    def __init__(self, n, F, device=None, t=None, ld=None):
        self.n, self.F = int(n), int(F)
        if t is None:
            t = torch.empty((self.n, int(ld) if ld else bf16_ld(F)), dtype=torch.bfloat16, device=device)
        assert t.dtype == torch.bfloat16 and t.is_contiguous() and t.dim() == 2 and t.shape[0] == self.n
        assert t.shape[1] >= pad4(F) and t.shape[1] % 4 == 0
        self.t = t

    @property
    def ld(self):
        return self.t.shape[1]

    @property
    def device(self):
        return self.t.device

    def numpy(self):
        return self.t[:, :self.F].float().cpu().numpy()

    def rows(self, r0, r1):
        """View of a row range (shares storage)."""
        return HMat(r1 - r0, self.F, t=self.t[r0:r1])


class Panels:
    """A row-partitioned matrix (n_rows x F) stored as W feature panels of width wp: t[q][i][j] = M[i][q*wp + j], each
    panel padded to R rows -- the send / receive layout of the multi-GPU feature repartition (one all-to-all hands panel
    q of every rank to rank q).  fp32, or bfloat16 (`bf16=True`: operand and wire format of the bf16 configuration).
    Pure container (device-agnostic, like DMat)."""
    __slots__ = ('t', 'n', 'F', 'R', 'W', 'wp', 'bf16')

    def __init__(self, n, F, R, W, wp, device, bf16=False, t=None):
        self.n, self.F, self.R, self.W, self.wp, self.bf16 = int(n), int(F), int(R), int(W), int(wp), bool(bf16)
        assert self.W * self.wp >= self.F and self.R >= self.n and self.wp % (8 if bf16 else 4) == 0
        if t is None:
            t = torch.zeros(self.W * self.R * self.wp, dtype=torch.bfloat16 if bf16 else torch.float32, device=device)
        self.t = t

    @property
    def device(self):
        return self.t.device

    def as_rows(self):
        """The buffer seen as ONE matrix of W*R rows and wp columns (what rank q's SpMM gathers after the exchange:
        row r*R + i = row i of rank r)."""
        v = self.t.view(self.W * self.R, self.wp)
        return HMat(self.W * self.R, self.wp, t=v) if self.bf16 else DMat(self.W * self.R, self.wp, t=v)


def bf16_ld(F):
    """Pitch (elements) of a bf16 gather operand: rows start on 128-byte lines."""
    return (int(F) + 63) // 64 * 64


def bf16_gather(precision=None):
    """True in the bf16 configuration ('bf16': BASELINE config 5): the SpMM then gathers a bf16 copy of its
    dense operand.  'f32' and 'bf16x3' keep the fp32 operand (their results are fp32-class)."""
    return (precision or GEMM_PRECISION) == 'bf16'


def cast_bf16(X: DMat, out: HMat = None):
    """fp32 -> bf16 (round to nearest even), whole pitch written."""
    out = HMat(X.n, X.F, X.device) if out is None else out
    check(_ffi.lib().geogcn_cast_bf16_f32(X.n, X.F, _p(X.t), X.ld, _p(out.t), out.ld, _stream()), 'cast_bf16_f32')
    return out


# Plans own device memory (hipFree in their destroy call).  A destructor may run at ANY time -- also while a training step
# is being captured into a hipGraph, where a free invalidates the capture ("operation failed due to a previous error
# during capture"; found by the random sweeps: an earlier model's CSR collected in the middle of a later model's
# capture).  So handles released during a capture are parked and destroyed at the next opportunity.
_parked = []


def _capturing():
    try:
        return torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()
    except Exception:
        return False


def _release(destroy, handle):
    if _capturing():
        _parked.append((destroy, handle))
        return
    while _parked:
        d, h = _parked.pop()
        d(h)
    destroy(handle)


class Workspace:
    """Grow-only scratch buffer (split-K slabs, long-row partials, reduction partials)."""

    def __init__(self, device):
        self.device = device
        self.t = torch.empty(0, dtype=torch.uint8, device=device)

    def get(self, nbytes):
        nbytes = int(nbytes)
        if self.t.numel() < nbytes:
            self.t = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=self.device)
        return self.t


class CSR:
    """Device CSR (int32 indices, fp32 values, rows in ascending column order) + the long-row split plan for the SpMM
    kernel."""

    def __init__(self, m: sps.spmatrix, device, long_row_nnz=None, chunk_nnz=None, sort=True, local=None):
        """`sort=False` keeps the given stored order inside each row (the SpMM accumulates in stored order and does not
        need it ascending; geogcn_xt_plan_create does and checks).  `local`: whether the row numbering has locality (a long
        row's neighbours lie near it) -- decides on which XCD the long rows' chunks run (geogcn_spmm_plan_create); None =
        measured here: at least half of the long rows' entries within tuning.L2_WINDOW_ROWS of their row."""
        require_gpu()
        m = sps.csr_matrix(m)
        if long_row_nnz is None:
            # a row is walked by ONE 16-lane group, two nonzeros per ~0.7 us round trip: on a small matrix (CMU
            # shape) a 256-nonzero row alone takes ~90 us, longer than everything else in the launch -- cut rows
            # earlier there; on a large one the other rows hide it and fewer partial sums are cheaper
            long_row_nnz, chunk_nnz = (256, chunk_nnz or 128) if m.nnz >= 2_000_000 else (48, chunk_nnz or 48)
        elif chunk_nnz is None:
            chunk_nnz = 128
        if sort and not m.has_sorted_indices:
            m = m.copy()
            m.sort_indices()
        if m.nnz >= 2 ** 31 or max(m.shape) >= 2 ** 31:
            raise ValueError("CSR too large for int32 indices")
        self.shape = m.shape
        self.nnz = int(m.nnz)
        indptr = np.ascontiguousarray(m.indptr, dtype=np.int32)
        indices = np.ascontiguousarray(m.indices, dtype=np.int32)
        data = np.ascontiguousarray(m.data, dtype=np.float32)
        self.rowptr_host = indptr
        self.colidx_host = indices          # (host copy: plan builders walk the structure, e.g. geogcn_xt_plan_create)
        self.rowptr = torch.from_numpy(indptr).to(device)
        self.colidx = torch.from_numpy(indices).to(device)
        self.val = torch.from_numpy(data).to(device)
        self.device = device
        self._plan = C.c_void_p(0)
        lib = _ffi.lib()
        if local is None:
            local = False
            deg = np.diff(indptr)
            long_rows = np.nonzero(deg > long_row_nnz)[0]
            if len(long_rows) and m.shape[0] == m.shape[1]:
                sel = np.concatenate([np.arange(indptr[r], indptr[r + 1]) for r in long_rows[:4096]])
                row_of = np.repeat(long_rows[:4096], deg[long_rows[:4096]])
                local = bool(np.mean(np.abs(indices[sel].astype(np.int64) - row_of) <= tuning.L2_WINDOW_ROWS) >= 0.5)
        self.chunks_with_owner = bool(local)
        check(lib.geogcn_spmm_plan_create(self.shape[0], indptr.ctypes.data_as(C.c_void_p),
                                          int(long_row_nnz), int(chunk_nnz), int(self.chunks_with_owner), C.byref(self._plan)),
              'spmm_plan_create')
        self.n_long_rows = int(lib.geogcn_spmm_plan_num_long_rows(self._plan))
        self.n_chunks = int(lib.geogcn_spmm_plan_num_chunks(self._plan))
        self._ws = Workspace(device)

    def __del__(self):
        try:
            if self._plan:
                _release(_ffi.lib().geogcn_spmm_plan_destroy, self._plan)
                self._plan = C.c_void_p(0)
        except Exception:
            pass


def spmm(A: CSR, B, out: DMat = None, bias: torch.Tensor = None, act=ACT_NONE, F=None, accumulate=False, out_col0=0):
    """out = act(A . B + bias)  -- S.structured_dot (reference gcnmodel.py:39,130,153).  B is a DMat, or an
    HMat (bf16 gathered operand, fp32 accumulation) in the bf16 configuration.  accumulate: out = act(out + A . B + bias).
    out_col0 (a multiple of 4): the F result columns go to columns [out_col0, out_col0 + F) of `out` (the feature-slab pipeline
    of the partitioned all-gather: one product per slab into one output matrix; bias is then that slab's slice)."""
    lib = _ffi.lib()
    F = B.F if F is None else F
    if out_col0:
        if out is None or out_col0 % 4 or out_col0 + pad4(F) > out.ld or accumulate:
            raise ValueError("spmm: out_col0 needs an existing output with room for the slab, a multiple of 4, no accumulate")
    if B.n != A.shape[1]:
        raise ValueError("spmm: A is %s but B has %d rows" % (A.shape, B.n))
    if out is None:
        if accumulate:
            raise ValueError("spmm: accumulate needs an existing output")
        out = DMat.empty(A.shape[0], F, B.device)
    need = lib.geogcn_spmm_workspace_bytes(A._plan, F)
    ws = A._ws.get(need)
    if accumulate and isinstance(B, HMat):
        raise ValueError("spmm: the accumulate form takes an fp32 operand")
    fn, name = ((lib.geogcn_spmm_csr_bf16b, 'spmm_csr_bf16b') if isinstance(B, HMat)
                else (lib.geogcn_spmm_csr_acc_f32, 'spmm_csr_acc_f32') if accumulate
                else (lib.geogcn_spmm_csr_f32, 'spmm_csr_f32'))
    cptr = _p(out.t) if not out_col0 else C.c_void_p(out.t.data_ptr() + 4 * int(out_col0))
    check(fn(A._plan, A.shape[0], A.shape[1], A.nnz, _p(A.rowptr), _p(A.colidx), _p(A.val), _p(B.t), B.ld,
             cptr, out.ld, F, _p(bias), act, _p(ws), ws.numel(), _stream()), name)
    return out


def spmm_softmax_ok(B, F):
    """Can softmax(A . B + bias) come out of the graph product's epilogue (geogcn_spmm_csr_softmax_f32)?  fp32 gathered operand,
    32 < F <= 512."""
    return isinstance(B, DMat) and 32 < int(F) <= 512


@_timed('spmm_softmax')
def spmm_softmax(A: CSR, B: DMat, bias: torch.Tensor = None, F=None, out: DMat = None, argmax: torch.Tensor = None):
    """out = softmax(A . B + bias) row by row, `argmax` (int32, optional) = the first index of each row's maximum -- the output
    layer's graph product with its nonlinearity in the epilogue; the logits are never written."""
    lib = _ffi.lib()
    F = B.F if F is None else F
    if B.n != A.shape[1]:
        raise ValueError("spmm_softmax: A is %s but B has %d rows" % (A.shape, B.n))
    out = DMat.empty(A.shape[0], F, B.device) if out is None else out
    ws = A._ws.get(lib.geogcn_spmm_workspace_bytes(A._plan, F))
    check(lib.geogcn_spmm_csr_softmax_f32(A._plan, A.shape[0], A.shape[1], A.nnz, _p(A.rowptr), _p(A.colidx), _p(A.val), _p(B.t), B.ld,
                                          _p(out.t), out.ld, F, _p(bias), _p(argmax), _p(ws), ws.numel(), _stream()), 'spmm_csr_softmax_f32')
    return out


@_timed('spmm_highway')
def spmm_highway(A: CSR, B, bias: torch.Tensor, T: DMat, H: DMat, Hc: DMat = None, Hout: DMat = None):
    """(Hc, Hout) = (tanh(A . B + bias), T*Hc + (1-T)*H) in one launch -- the highway block's convolution with
    the gating mix (reference gcnmodel.py:266) fused into the SpMM's epilogue."""
    lib = _ffi.lib()
    F = H.F
    if B.n != A.shape[1] or T.n != A.shape[0] or H.n != A.shape[0] or T.ld != H.ld:
        raise ValueError("spmm_highway: shapes / pitches do not match")
    Hc = DMat.empty(H.n, F, H.device, ld=H.ld) if Hc is None else Hc
    Hout = DMat.empty(H.n, F, H.device, ld=H.ld) if Hout is None else Hout
    if Hc.ld != H.ld or Hout.ld != H.ld:
        raise ValueError("spmm_highway: T, H, Hc, Hout must share one pitch")
    need = lib.geogcn_spmm_workspace_bytes(A._plan, F)
    ws = A._ws.get(need)
    check(lib.geogcn_spmm_csr_highway_f32(A._plan, A.shape[0], A.shape[1], A.nnz, _p(A.rowptr), _p(A.colidx), _p(A.val),
                                          _p(B.t), B.ld, int(isinstance(B, HMat)), F, _p(bias), _p(T.t), _p(H.t), H.ld,
                                          _p(Hc.t), _p(Hout.t), _p(ws), ws.numel(), _stream()), 'spmm_csr_highway_f32')
    return Hc, Hout


_gemm_ws = {}

# How the activation x weight products are formed (include/geogcn.h GEOGCN_GEMM_*): 'f32' = exact fp32 MFMA everywhere,
# 'bf16x3' (default since round 5) = three-term bf16 split with fp32 accumulation (fp32-class accuracy) on the kernels of
# csrc/gemm_x3.hip where they take the shape -- the TwitterUS-size products -- and exact fp32 elsewhere, 'bf16' = BASELINE config 5.
GEMM_PRECISIONS = {'f32': _ffi.GEMM_F32, 'bf16x3': _ffi.GEMM_BF16X3, 'bf16': _ffi.GEMM_BF16}
GEMM_PRECISION = tuning.GEMM_PRECISION


def _gemm_label(A, B, out=None, transA=False, transB=False, bias=None, act=ACT_NONE, accumulate=False, precision=None, gate_carry=None):
    # 'gemm:<form>:<precision>:<M>:<N>:<K>:<bytes per element of C>:<accumulate>' (bench.py prices the entry from it)
    M, K = (A.F, A.n) if transA else (A.n, A.F)
    N = B.n if transB else B.F
    return 'gemm:%s:%s:%d:%d:%d:%d:%d' % ('tn' if transA else 'nt' if transB else 'nn', precision or GEMM_PRECISION, M, N, K,
                                          2 if isinstance(out, HMat) or (isinstance(out, Panels) and out.bf16) else 4, int(bool(accumulate)))


@_timed(_gemm_label)
def gemm(A: DMat, B: DMat, out: DMat = None, transA=False, transB=False, bias=None, act=ACT_NONE,
         accumulate=False, precision=None, gate_carry=None):
    """out = act(op(A) . op(B) + bias) [+ out]  -- T.dot / Gemm (reference gcnmodel.py:126,149,285).
    `gate_carry` (a GateCarry): out = A . op(B) + G * (1 - T), the highway block's carry gradient formed in the epilogue
    (geogcn_gemm_gated_f32; no bias / activation / accumulate / transA)."""
    lib = _ffi.lib()
    M = A.F if transA else A.n
    K = A.n if transA else A.F
    N = B.n if transB else B.F
    Kb = B.F if transB else B.n
    if Kb != K:
        raise ValueError("gemm: inner dimensions differ (%d vs %d)" % (K, Kb))
    if out is None:
        out = DMat.empty(M, N, A.device)
    dev = A.device
    ws = _gemm_ws.get(dev)
    if ws is None:
        ws = _gemm_ws[dev] = Workspace(dev)
    if isinstance(out, Panels):
        # multi-GPU: the product goes straight into the all-to-all's send layout (feature panels)
        if transA or accumulate:
            raise ValueError("gemm: a panel result needs transA=False, accumulate=False")
        prec = _ffi.GEMM_BF16 if out.bf16 else GEMM_PRECISIONS[precision or GEMM_PRECISION]
        w = ws.get(lib.geogcn_gemm_workspace_bytes(0, int(transB), M, N, K, prec))
        check(lib.geogcn_gemm_panels_f32(int(transB), M, N, K, _p(A.t), A.ld, _p(B.t), B.ld, _p(out.t), out.R, out.W, out.wp,
                                         int(out.bf16), _p(bias), act, prec, _p(w), w.numel(), _stream()), 'gemm_panels_f32')
        return out
    if isinstance(out, HMat):
        # bf16 configuration: the product lands in bf16, the SpMM's operand format (no fp32 round trip)
        if transA or accumulate or (precision or GEMM_PRECISION) != 'bf16':
            raise ValueError("gemm: a bf16 result needs transA=False, accumulate=False, precision='bf16'")
        w = ws.get(lib.geogcn_gemm_workspace_bytes(0, int(transB), M, N, K, _ffi.GEMM_BF16))
        check(lib.geogcn_gemm_f32_bf16c(int(transB), M, N, K, _p(A.t), A.ld, _p(B.t), B.ld, _p(out.t), out.ld,
                                        _p(bias), act, _p(w), w.numel(), _stream()), 'gemm_f32_bf16c')
        return out
    prec = GEMM_PRECISIONS[precision or GEMM_PRECISION]
    need = lib.geogcn_gemm_workspace_bytes(int(transA), int(transB), M, N, K, prec)
    w = ws.get(need)
    if gate_carry is not None:
        g, t = gate_carry.G, gate_carry.T
        if transA or accumulate or bias is not None or act != ACT_NONE:
            raise ValueError("gemm: a gate carry excludes transA, accumulate, bias and activation")
        if g.n != M or g.F != N or t.n != M or t.F != N:
            raise ValueError("gemm: the gate carry's operands do not have the output's shape")
        check(lib.geogcn_gemm_gated_f32(int(transB), M, N, K, _p(A.t), A.ld, _p(B.t), B.ld, _p(out.t), out.ld, _p(g.t), g.ld,
                                        _p(t.t), t.ld, prec, _p(w), w.numel(), _stream()), 'gemm_gated_f32')
        return out
    check(lib.geogcn_gemm_f32(int(transA), int(transB), M, N, K, _p(A.t), A.ld, _p(B.t), B.ld, _p(out.t),
                              out.ld, _p(bias), act, int(accumulate), prec, _p(w), w.numel(), _stream()),
          'gemm_f32')
    return out


def _fused_precision(precision, allow_bf16=False):
    """The fused highway launches exist in two precisions: 'f32' and 'bf16x3'; the k-concatenated product (gemm_kcat) also in 'bf16'
    (round 6: one launch of the bf16 whole-rows kernel; the bf16 configuration keeps its own launches for the other pairs)."""
    p = precision or GEMM_PRECISION
    if p not in ('f32', 'bf16x3') and not (allow_bf16 and p == 'bf16'):
        raise ValueError("fused highway GEMM launches take precision 'f32' or 'bf16x3'%s, not %r" % (" or 'bf16'" if allow_bf16 else '', p))
    return GEMM_PRECISIONS[p]


@_timed(lambda A, B0, B1, **kw: 'gemm_dual_tn' if kw.get('transA') else 'gemm_dual_nn')
def gemm_dual(A: DMat, B0: DMat, B1: DMat, out0: DMat = None, out1: DMat = None, transA=False, bias0=None, act0=ACT_NONE,
              bias1=None, act1=ACT_NONE, precision=None):
    """(out0, out1) = (act0(op(A) . B0 + bias0), act1(op(A) . B1 + bias1)) in ONE launch: the highway block's conv branch and
    gate read the same input (reference gcnmodel.py:281-286).  `precision`: 'f32' (exact) or 'bf16x3' (fp32-class split-bf16
    products where a kernel takes the shape, exact otherwise)."""
    lib = _ffi.lib()
    M = A.F if transA else A.n
    K = A.n if transA else A.F
    if B0.n != K or B1.n != K:
        raise ValueError("gemm_dual: inner dimensions differ (%d vs %d, %d)" % (K, B0.n, B1.n))
    out0 = DMat.empty(M, B0.F, A.device) if out0 is None else out0
    out1 = DMat.empty(M, B1.F, A.device) if out1 is None else out1
    dev = A.device
    ws = _gemm_ws.get(dev)
    if ws is None:
        ws = _gemm_ws[dev] = Workspace(dev)
    prec = _fused_precision(precision, allow_bf16=bool(transA))          # (bf16: the two weight gradients; its forward pair is gemm_dual_bf16)
    w = ws.get(lib.geogcn_gemm_dual_workspace_bytes(int(transA), M, B0.F, B1.F, K, prec))
    check(lib.geogcn_gemm_dual_f32(int(transA), M, B0.F, B1.F, K, _p(A.t), A.ld, _p(B0.t), B0.ld, _p(B1.t), B1.ld,
                                   _p(out0.t), out0.ld, _p(out1.t), out1.ld, _p(bias0), int(act0), _p(bias1), int(act1),
                                   prec, _p(w), w.numel(), _stream()), 'gemm_dual_f32')
    return out0, out1


class GateCarry:
    """The carry gradient of a highway block, G * (1 - T), NOT yet formed: the gating layer hands it down like a gradient matrix and
    gemm_kcat(gate_carry=...) forms it in its epilogue (geogcn_gemm_kcat_gated_f32) -- highway_bwd then does not store it and
    nothing reads it back.  Whoever cannot take it that way calls `dense()`."""
    __slots__ = ('G', 'T')

    def __init__(self, G: DMat, T: DMat):
        self.G, self.T = G, T

    def dense(self):
        out = self.G.like()
        check(_ffi.lib().geogcn_gate_carry_f32(self.G.n, self.G.F, _p(self.G.t), self.G.ld, _p(self.T.t), self.T.ld, _p(out.t),
                                               out.ld, _stream()), 'gate_carry_f32')
        return out


def gemm_gated_native(n, F, precision=None):
    """Does ONE product dH = dZ . W^T + G * (1 - T) (the separate launches of the reverse sweep: bf16 configuration, FUSE_GEMMS
    off) form the carry in its epilogue?  bf16: the whole-rows kernel takes every GCN width (any other falls back to carry +
    accumulate at the cost of the stored carry); fp32: from 32,768 rows on."""
    p = precision or GEMM_PRECISION
    if p == 'bf16':
        return True
    lib = _ffi.lib()
    if p == 'bf16x3' and (lib.geogcn_gemm_workspace_bytes(0, 1, int(n), int(F), int(F), _ffi.GEMM_BF16X3)
                          > 3 * ((int(F) + 31) // 32 * 32) * int(F) * 2):
        return True          # the split-bf16 whole-rows kernel takes it (its fragment-ordered weights are larger than the staged kernel's planes)
    return p in ('f32', 'bf16x3') and lib.geogcn_gemm_workspace_bytes(0, 1, int(n), int(F), int(F), _ffi.GEMM_F32) > 0


def kcat_gated_native(n, F, precision=None):
    """Does dH = dZ . Wh^T + dU . Wt^T + G * (1 - T) run with the carry in the epilogue at this size (a whole-rows kernel)?"""
    lib = _ffi.lib()
    if (precision or GEMM_PRECISION) == 'bf16':
        # the k-concatenated bf16 launch holds the fragments of BOTH weights: a larger workspace than a single product's says it is taken
        return (tuning.FUSE_BF16_KCAT and lib.geogcn_gemm_kcat_workspace_bytes(1, int(n), int(F), int(F), int(F), _ffi.GEMM_BF16)
                > lib.geogcn_gemm_workspace_bytes(0, 1, int(n), int(F), int(F), _ffi.GEMM_BF16))
    return lib.geogcn_gemm_kcat_workspace_bytes(1, int(n), int(F), int(F), int(F), _fused_precision(precision)) > 0


@_timed('gemm_dual_bf16')
def gemm_dual_bf16(A: DMat, B0: DMat, B1: DMat, out0=None, out1: DMat = None, bias1=None, act1=ACT_NONE):
    """(out0, out1) = (A . B0, act1(A . B1 + bias1)) with bf16 products in ONE launch (geogcn_gemm_dual_bf16): the bf16
    configuration's highway block -- out0 an HMat (bf16: what the SpMM gathers) or a DMat, out1 fp32."""
    lib = _ffi.lib()
    if B0.n != A.F or B1.n != A.F:
        raise ValueError("gemm_dual_bf16: inner dimensions differ (%d vs %d, %d)" % (A.F, B0.n, B1.n))
    out0 = HMat(A.n, B0.F, A.device) if out0 is None else out0
    out1 = DMat.empty(A.n, B1.F, A.device) if out1 is None else out1
    ws = _gemm_ws.get(A.device)
    if ws is None:
        ws = _gemm_ws[A.device] = Workspace(A.device)
    w = ws.get(lib.geogcn_gemm_dual_bf16_workspace_bytes(A.n, B0.F, B1.F, A.F))
    check(lib.geogcn_gemm_dual_bf16(A.n, B0.F, B1.F, A.F, _p(A.t), A.ld, _p(B0.t), B0.ld, _p(B1.t), B1.ld, _p(out0.t), out0.ld,
                                    int(isinstance(out0, HMat)), _p(out1.t), out1.ld, _p(bias1), int(act1), _p(w), w.numel(),
                                    _stream()), 'gemm_dual_bf16')
    return out0, out1


@_timed('gemm_kcat')
def gemm_kcat(A0: DMat, B0: DMat, A1: DMat, B1: DMat, out: DMat = None, transB=False, accumulate=False, gate_carry: GateCarry = None,
              tanh_bwd=None, precision=None):
    """out = A0 . op(B0) + A1 . op(B1) [+ out]: two products, one accumulator, one pass over out (`precision` 'f32', 'bf16x3' or 'bf16') --
    dH = dZ . Wh^T + dU . Wt^T of the highway block.  `gate_carry`: ... + G * (1 - T) formed in the epilogue (then no accumulate).
    `tanh_bwd` = (Y, keep_mask, scale), with a gate carry only: the result times keep * scale * (1 - Y^2) -- the dropout + tanh gradient
    of the layer below the first block in the same epilogue (geogcn_gemm_kcat_gated_tanhbwd_f32)."""
    N = B0.n if transB else B0.F
    if (B1.n if transB else B1.F) != N or A0.n != A1.n or (B0.F if transB else B0.n) != A0.F or (B1.F if transB else B1.n) != A1.F:
        raise ValueError("gemm_kcat: shapes do not match")
    if gate_carry is not None and accumulate:
        raise ValueError("gemm_kcat: a gate carry and accumulate exclude each other")
    if tanh_bwd is not None and gate_carry is None:
        raise ValueError("gemm_kcat: tanh_bwd comes with a gate carry")
    if out is None:
        if accumulate:
            raise ValueError("gemm_kcat: accumulate needs an existing output")
        out = DMat.empty(A0.n, N, A0.device)
    lib = _ffi.lib()
    ws = _gemm_ws.get(A0.device)
    if ws is None:
        ws = _gemm_ws[A0.device] = Workspace(A0.device)
    prec = _fused_precision(precision, allow_bf16=True)
    w = ws.get(lib.geogcn_gemm_kcat_workspace_bytes(int(transB), A0.n, N, A0.F, A1.F, prec))
    if gate_carry is not None:
        g, t = gate_carry.G, gate_carry.T
        if g.n != A0.n or g.F != N or t.n != A0.n or t.F != N:
            raise ValueError("gemm_kcat: the gate carry's operands do not have the output's shape")
        if tanh_bwd is not None:
            Y, keep, scale = tanh_bwd
            if Y.n != A0.n or Y.F != N or tuple(keep.shape) != (A0.n, N) or keep.dtype != torch.uint8 or not keep.is_contiguous() or N % 4:
                raise ValueError("gemm_kcat: tanh_bwd needs Y and a contiguous uint8 keep mask of the output's shape, width % 4 == 0")
            check(lib.geogcn_gemm_kcat_gated_tanhbwd_f32(int(transB), A0.n, N, A0.F, A1.F, _p(A0.t), A0.ld, _p(B0.t), B0.ld, _p(A1.t), A1.ld,
                                                         _p(B1.t), B1.ld, _p(out.t), out.ld, _p(g.t), g.ld, _p(t.t), t.ld, _p(Y.t), Y.ld,
                                                         _p(keep), N, float(scale), prec, _p(w), w.numel(), _stream()),
                  'gemm_kcat_gated_tanhbwd_f32')
            return out
        check(lib.geogcn_gemm_kcat_gated_f32(int(transB), A0.n, N, A0.F, A1.F, _p(A0.t), A0.ld, _p(B0.t), B0.ld, _p(A1.t), A1.ld,
                                             _p(B1.t), B1.ld, _p(out.t), out.ld, _p(g.t), g.ld, _p(t.t), t.ld, prec, _p(w), w.numel(),
                                             _stream()), 'gemm_kcat_gated_f32')
        return out
    check(lib.geogcn_gemm_kcat_f32(int(transB), A0.n, N, A0.F, A1.F, _p(A0.t), A0.ld, _p(B0.t), B0.ld, _p(A1.t), A1.ld,
                                   _p(B1.t), B1.ld, _p(out.t), out.ld, int(accumulate), prec, _p(w), w.numel(), _stream()), 'gemm_kcat_f32')
    return out


def bias_act(X: DMat, bias, act, out: DMat = None):
    out = X.like() if out is None else out
    check(_ffi.lib().geogcn_bias_act_f32(X.n, X.F, _p(X.t), X.ld, _p(bias), act, _p(out.t), out.ld,
                                         _stream()), 'bias_act_f32')
    return out


def highway_fwd(T: DMat, Hc: DMat, H: DMat, out: DMat = None):
    out = H.like() if out is None else out
    check(_ffi.lib().geogcn_highway_fwd_f32(H.n, H.F, _p(T.t), _p(Hc.t), _p(H.t), H.ld, _p(out.t),
                                            _stream()), 'highway_fwd_f32')
    return out


def highway_bwd_bf16_ok(G: DMat, with_bias: bool):
    """Can highway_bwd store dS as bfloat16 for this gradient?  (fused column sums: plain pitch, F <= 1024)"""
    return G.F <= 1024 and (not with_bias or G.ld == pad4(G.F))


def highway_bwd(G: DMat, T: DMat, Hc: DMat, H: DMat, dS: DMat = None, dU: DMat = None, dHcarry: DMat = None,
                dbS: torch.Tensor = None, dbU: torch.Tensor = None, dS_bf16=False, carry=True):
    """-> (dS, dU, dHcarry); with dbS / dbU given, also the two bias gradients (column sums) in the same pass.
    `dS_bf16`: dS comes back as an HMat (bf16, the bits cast_bf16 would produce) -- the bf16 configuration's A^T . dS gathers
    it directly, the fp32 dS and the cast pass are never written."""
    lib = _ffi.lib()
    dU = G.like() if dU is None else dU
    # (`carry=False`: dHcarry is not stored -- a GateCarry hands it to gemm_kcat / gemm; None comes back)
    dHcarry = (G.like() if dHcarry is None else dHcarry) if carry else None
    carry_t = dHcarry.t if carry else None
    w = _ws_for(G.device).get(max(lib.geogcn_highway_bwd_workspace_bytes(G.n, G.F),
                                  lib.geogcn_colsum_workspace_bytes(G.n, G.F)) if dbS is not None else 0)
    if dS_bf16:
        dS = HMat(G.n, G.F, G.device) if dS is None else dS
        check(lib.geogcn_highway_bwd_bf16s_f32(G.n, G.F, _p(G.t), _p(T.t), _p(Hc.t), _p(H.t), G.ld, _p(dS.t), dS.ld, _p(dU.t),
                                               _p(carry_t), _p(dbS), _p(dbU), _p(w), w.numel(), _stream()),
              'highway_bwd_bf16s_f32')
        return dS, dU, dHcarry
    dS = DMat.empty(G.n, G.F, G.device, ld=gather_ld(G.F)) if dS is None else dS      # dS feeds the A^T SpMM
    check(lib.geogcn_highway_bwd_f32(G.n, G.F, _p(G.t), _p(T.t), _p(Hc.t), _p(H.t), G.ld, _p(dS.t), dS.ld, _p(dU.t),
                                     _p(carry_t), _p(dbS), _p(dbU), _p(w), w.numel(), _stream()), 'highway_bwd_f32')
    return dS, dU, dHcarry


def act_bwd(G: DMat, Y: DMat, act, out: DMat = None, keep_mask=None, scale=1.0):
    """dS = G [* mask * scale] * act'(Y)  (act' through the layer output Y)."""
    out = G.like() if out is None else out
    check(_ffi.lib().geogcn_act_bwd_f32(G.n, G.F, _p(G.t), _p(Y.t), G.ld, int(act), _p(keep_mask), float(scale),
                                        _p(out.t), out.ld, _stream()), 'act_bwd_f32')
    return out


def act_bwd_colsum(G: DMat, Y: DMat, act, db: torch.Tensor, out: DMat = None, keep_mask=None, scale=1.0):
    """act_bwd plus the bias gradient db = column sums of dS in the same pass."""
    lib = _ffi.lib()
    out = G.like() if out is None else out
    w = _ws_for(G.device).get(max(lib.geogcn_highway_bwd_workspace_bytes(G.n, G.F),
                                  lib.geogcn_colsum_workspace_bytes(G.n, G.F)))
    check(lib.geogcn_act_bwd_colsum_f32(G.n, G.F, _p(G.t), _p(Y.t), G.ld, int(act), _p(keep_mask), float(scale),
                                        _p(out.t), out.ld, _p(db), _p(w), w.numel(), _stream()), 'act_bwd_colsum_f32')
    return out


def tanh_bwd(G: DMat, Y: DMat, out: DMat = None, keep_mask=None, scale=1.0):
    return act_bwd(G, Y, ACT_TANH, out, keep_mask, scale)


def add_inplace(X: DMat, Y: DMat):
    """Y += X (same shape / pitch)."""
    assert X.t.shape == Y.t.shape
    check(_ffi.lib().geogcn_add_inplace_f32(X.t.numel(), _p(X.t), _p(Y.t), _stream()), 'add_inplace_f32')
    return Y


_misc_ws = {}


def _ws_for(dev):
    ws = _misc_ws.get(dev)
    if ws is None:
        ws = _misc_ws[dev] = Workspace(dev)
    return ws


def colsum_rowblocks(X: DMat, out: torch.Tensor):
    """Column sums in the order of the fused activation-gradient kernels (geogcn_colsum_rowblocks_f32): a bias gradient taken from
    a dS that a product's epilogue wrote equals, bit for bit, the one act_bwd_colsum would have produced."""
    lib = _ffi.lib()
    w = _ws_for(X.device).get(max(lib.geogcn_highway_bwd_workspace_bytes(X.n, X.F), lib.geogcn_colsum_workspace_bytes(X.n, X.F)))
    check(lib.geogcn_colsum_rowblocks_f32(X.n, X.F, _p(X.t), X.ld, _p(out), _p(w), w.numel(), _stream()), 'colsum_rowblocks_f32')
    return out


def colsum(X: DMat, out: torch.Tensor = None):
    lib = _ffi.lib()
    if out is None:
        out = torch.zeros(pad4(X.F), dtype=torch.float32, device=X.device)
    w = _ws_for(X.device).get(lib.geogcn_colsum_workspace_bytes(X.n, X.F))
    check(lib.geogcn_colsum_f32(X.n, X.F, _p(X.t), X.ld, _p(out), _p(w), w.numel(), _stream()), 'colsum_f32')
    return out


def dropout_mask(n, F, p, seed, offset, device, out=None):
    if out is None:
        out = torch.empty((n, F), dtype=torch.uint8, device=device)
    check(_ffi.lib().geogcn_dropout_mask_philox(n, F, float(p), int(seed), int(offset), _p(out), _stream()),
          'dropout_mask_philox')
    return out


def dropout_mask_ctr(n, F, p, seed, calls_dev, per_call_elems, base_elems, device, out=None):
    """Same stream as dropout_mask, positioned by a device-resident call counter (captured steps)."""
    if out is None:
        out = torch.empty((n, F), dtype=torch.uint8, device=device)
    check(_ffi.lib().geogcn_dropout_mask_philox_ctr(n, F, float(p), int(seed), _p(calls_dev), int(per_call_elems),
                                                    int(base_elems), _p(out), _stream()), 'dropout_mask_philox_ctr')
    return out


def counter_add(counter_dev, delta=1):
    check(_ffi.lib().geogcn_counter_add_i64(_p(counter_dev), int(delta), _stream()), 'counter_add_i64')


def dropout_apply(X: DMat, keep_mask, p, out: DMat = None):
    out = X.like() if out is None else out
    check(_ffi.lib().geogcn_dropout_apply_f32(X.n, X.F, _p(X.t), X.ld, _p(keep_mask), float(p), _p(out.t),
                                              _stream()), 'dropout_apply_f32')
    return out


def softmax_rows(L: DMat, out: DMat = None, argmax: torch.Tensor = None):
    out = L.like() if out is None else out
    check(_ffi.lib().geogcn_softmax_rows_f32(L.n, L.F, _p(L.t), L.ld, _p(out.t), out.ld, _p(argmax),
                                             _stream()), 'softmax_rows_f32')
    return out


def ce_metrics(P: DMat, idx: torch.Tensor, y: torch.Tensor, argmax: torch.Tensor = None, out2=None):
    """-> device tensor [sum of -log P[idx, y], number of argmax hits] (gcnmodel.py:376-382)."""
    lib = _ffi.lib()
    if out2 is None:
        out2 = torch.zeros(2, dtype=torch.float32, device=P.device)
    w = _ws_for(P.device).get(lib.geogcn_ce_metrics_workspace_bytes(idx.numel()))
    check(lib.geogcn_ce_metrics_f32(P.F, _p(P.t), P.ld, _p(argmax), _p(idx), idx.numel(), _p(y), _p(out2),
                                    _p(w), w.numel(), _stream()), 'ce_metrics_f32')
    return out2


def softmax_ce_bwd(P: DMat, idx: torch.Tensor, y: torch.Tensor, out: DMat = None, inv_n=None, db: torch.Tensor = None):
    """dlogits of the mean cross-entropy over the indexed rows; with `db` also its column sums (bias gradient)."""
    out = P.like() if out is None else out
    if inv_n is None:
        inv_n = 1.0 / max(1, idx.numel())
    if db is not None:
        lib = _ffi.lib()
        w = _ws_for(P.device).get(lib.geogcn_softmax_ce_bwd_db_workspace_bytes(P.F))
        check(lib.geogcn_softmax_ce_bwd_db_f32(P.n, P.F, _p(P.t), P.ld, _p(idx), idx.numel(), _p(y), float(inv_n),
                                               _p(out.t), out.ld, _p(db), _p(w), w.numel(), _stream()),
              'softmax_ce_bwd_db_f32')
        return out
    check(_ffi.lib().geogcn_softmax_ce_bwd_f32(P.n, P.F, _p(P.t), P.ld, _p(idx), idx.numel(), _p(y), float(inv_n),
                                               _p(out.t), out.ld, _stream()), 'softmax_ce_bwd_f32')
    return out


def softmax_ce_rows_bwd(P: DMat, idx: torch.Tensor, y: torch.Tensor, inv_n, db: torch.Tensor, out: DMat = None):
    """COMPACT gradient of the mean cross-entropy: one row per index (row j belongs to output row idx[j]) + the bias
    gradient db; geogcn_softmax_ce_rows_bwd_db_f32."""
    lib = _ffi.lib()
    out = DMat.empty(idx.numel(), P.F, P.device, ld=gather_ld(P.F)) if out is None else out
    w = _ws_for(P.device).get(lib.geogcn_softmax_ce_bwd_db_workspace_bytes(P.F))
    check(lib.geogcn_softmax_ce_rows_bwd_db_f32(P.F, _p(P.t), P.ld, _p(idx), idx.numel(), _p(y), float(inv_n), _p(out.t), out.ld,
                                                _p(db), _p(w), w.numel(), _stream()), 'softmax_ce_rows_bwd_db_f32')
    return out


def gather_rows(X: DMat, idx: torch.Tensor, out: torch.Tensor = None):
    if out is None:
        out = torch.empty((idx.numel(), X.F), dtype=torch.float32, device=X.device)
    check(_ffi.lib().geogcn_gather_rows_f32(X.F, _p(X.t), X.ld, _p(idx), idx.numel(), _p(out), out.shape[1],
                                            _stream()), 'gather_rows_f32')
    return out


def pack_rows(X, idx: torch.Tensor, out: torch.Tensor):
    """out[j] = the whole PITCHED row idx[j] of X (DMat or HMat; `out` has X's dtype and pitch): the halo exchange's send
    buffer (dist.py).  bf16 rows travel as pairs -- the pitch is even -- through the same 16-byte-per-lane kernel."""
    if idx.numel() == 0:
        return out
    assert out.dtype == X.t.dtype and out.shape[1] == X.ld and out.is_contiguous()
    src, dst = (X.t.view(torch.float32), out.view(torch.float32)) if X.t.dtype == torch.bfloat16 else (X.t, out)
    w = src.shape[1]
    check(_ffi.lib().geogcn_gather_rows_f32(w, _p(src), w, _p(idx), idx.numel(), _p(dst), w, _stream()), 'gather_rows_f32')
    return out


def scatter_rows(src: DMat, idx: torch.Tensor, out: DMat):
    """out[idx[j], :] = src[j, :]"""
    check(_ffi.lib().geogcn_scatter_rows_f32(src.F, _p(src.t), src.ld, _p(idx), idx.numel(), _p(out.t), out.ld,
                                             _stream()), 'scatter_rows_f32')
    return out


def pack_panels(X: DMat, R: int, W: int, wp: int, out: torch.Tensor):
    """out[q][i][j] = X[i][q*wp + j]  (row-partitioned -> per-destination feature panels)."""
    check(_ffi.lib().geogcn_pack_panels_f32(X.n, int(R), X.F, _p(X.t), X.ld, int(W), int(wp), _p(out), _stream()),
          'pack_panels_f32')
    return out


def cast_bf16_flat(src: torch.Tensor, dst: torch.Tensor, width: int):
    """fp32 -> bf16 over a flat buffer seen as rows of `width` elements (panel buffers)."""
    n = src.numel() // width
    check(_ffi.lib().geogcn_cast_bf16_f32(n, width, _p(src), width, _p(dst), width, _stream()), 'cast_bf16_f32')
    return dst


def unpack_panels(inp: torch.Tensor, R: int, W: int, wp: int, out: DMat):
    """out[i][q*wp + j] = inp[q][i][j]"""
    check(_ffi.lib().geogcn_unpack_panels_f32(out.n, int(R), out.F, _p(inp), int(W), int(wp), _p(out.t), out.ld,
                                              _stream()), 'unpack_panels_f32')
    return out


def adam_step(p, g, m, v, regmask, lr, b1, b2, eps, t, l1=0.0, l2=0.0):
    check(_ffi.lib().geogcn_adam_step_f32(p.numel(), _p(p), _p(g), _p(m), _p(v), _p(regmask), lr, b1, b2, eps,
                                          int(t), float(l1), float(l2), _stream()), 'adam_step_f32')


def adam_step_ctr(p, g, m, v, regmask, lr, b1, b2, eps, state_dev, l1=0.0, l2=0.0):
    """adam_step with the step index on the device (state_dev: int64[2], see geogcn.h): graph-capturable."""
    check(_ffi.lib().geogcn_adam_step_ctr_f32(p.numel(), _p(p), _p(g), _p(m), _p(v), _p(regmask), lr, b1, b2, eps,
                                              _p(state_dev), float(l1), float(l2), _stream()), 'adam_step_ctr_f32')


def reg_penalty(p, regmask, l1, l2, out=None):
    if out is None:
        out = torch.zeros(1, dtype=torch.float32, device=p.device)
    w = _ws_for(p.device).get(4096)
    check(_ffi.lib().geogcn_reg_penalty_f32(p.numel(), _p(p), _p(regmask), float(l1), float(l2), _p(out), _p(w),
                                            w.numel(), _stream()), 'reg_penalty_f32')
    return out


# (thresholds of the X path -- dense head panel, document-blocked X^T sweep, hot rows in LDS -- live in tuning.py)


class SparseOperand:
    """A constant sparse matrix as the path uses it: ``fwd`` multiplies it (A . B) and ``bwd``
    multiplies its transpose (A^T . G, the StructuredDot gradient).  For the normalised adjacency
    of an unweighted graph A^T == A exactly in fp32 (checked here, not assumed), so
    one CSR serves both directions; otherwise CSR(A^T) is built once on the host.

    For a bag-of-words X (Zipfian columns) the TRANSPOSED product is split: the few columns denser than
    the cost model of dense_head_size picks form a dense N x K panel (``head_dense``) whose X^T . G runs on the MFMA pipe (split-K GEMM),
    and the long tail stays sparse: ``bwd`` holds the tail rows of X^T (head rows empty).  The forward product X . W
    serves the hot rows of W from LDS instead (HotCSR)."""

    def __init__(self, fwd: CSR, bwd: CSR, symmetric: bool, head_idx=None, head_dense=None):
        self.fwd, self.bwd, self.symmetric = fwd, bwd, symmetric
        self.head_idx, self.head_dense = head_idx, head_dense
        self.shape = fwd.shape
        self._xt_plans = {}
        self._hot = {}            # HotCSR per LDS capacity (forward X . W0 with the hot rows of W0 in LDS)

    def xt_plan(self, F):
        """Plan of the document-blocked X^T . G kernel for width F (built on first use; None when the tail is small)."""
        if self.bwd is None or self.symmetric or self.bwd.nnz < tuning.XT_MIN_NNZ:
            return None
        base = getattr(self.bwd, '_base', self.bwd)           # value-dropout variants share the structure's plans
        plans = base._xt_plans if hasattr(base, '_xt_plans') else self._xt_plans
        plan = plans.get(int(F))
        if plan is None:
            plan = plans[int(F)] = XtPlan(base, F)
        return plan

    @staticmethod
    def from_scipy(m, device, need_transpose=True, long_row_nnz=None, chunk_nnz=None, dense_head=True):
        m = sps.csr_matrix(m).astype(np.float32)
        m.sort_indices()
        fwd = CSR(m, device, long_row_nnz, chunk_nnz)
        if not need_transpose:
            return SparseOperand(fwd, None, False)
        sym = False
        mt = sps.csr_matrix(m.T)
        mt.sort_indices()
        if m.shape[0] == m.shape[1]:
            sym = (np.array_equal(mt.indptr, m.indptr) and np.array_equal(mt.indices, m.indices)
                   and np.array_equal(mt.data, m.data))
        if sym:
            return SparseOperand(fwd, fwd, True)
        head_idx = head_dense = None
        if dense_head and m.shape[0] > 0:
            col_nnz = np.diff(mt.indptr)
            k = dense_head_size(col_nnz, m.shape[0])
            cand = np.sort(np.argsort(-col_nnz, kind='stable')[:k])
            if len(cand) >= 16:
                panel = DMat(m.shape[0], len(cand), device)
                panel.t[:, :len(cand)].copy_(torch.from_numpy(np.ascontiguousarray(m[:, cand].toarray())))
                head_dense = panel
                head_idx = torch.from_numpy(cand.astype(np.int32)).to(device)
                keep = np.ones(mt.shape[0], dtype=bool)
                keep[cand] = False
                mt = sps.diags(keep.astype(np.float32)).tocsr() @ mt     # empty the head rows of X^T
                mt = sps.csr_matrix(mt)
                mt.eliminate_zeros()
                mt.sort_indices()
        bwd = CSR(mt, device, long_row_nnz, chunk_nnz)
        bwd._xt_plans = {}
        return SparseOperand(fwd, bwd, False, head_idx, head_dense)


def dense_head_size(col_nnz, n_rows):
    """How many of the densest columns of a bag-of-words X go into the dense head panel of X^T . G (0 = none): the size among
    tuning.DENSE_HEAD_SIZES (whole GEMM tiles) that minimises  padded rows x 2 N / GEMM rate  +  tail entries x 4 B / gather
    rate  (per output column; both terms scale with the output width alike)."""
    nnz_sorted = np.sort(np.asarray(col_nnz, dtype=np.int64))[::-1]
    nonempty = int(np.count_nonzero(nnz_sorted))
    cum = np.concatenate([[0], np.cumsum(nnz_sorted)])
    total = int(cum[-1])

    def cost(k):
        padded = 0 if k == 0 else min(-(-k // 128) * 128, -(-k // 160) * 160)
        return padded * 2.0 * n_rows / tuning.DENSE_HEAD_GEMM_FLOPS + (total - int(cum[k])) * 4.0 / tuning.DENSE_HEAD_GATHER_BYTES_PER_S
    sizes = [k for k in tuning.DENSE_HEAD_SIZES if k <= min(nonempty, tuning.DENSE_HEAD_MAX_COLS)]
    best = min([0] + sizes, key=cost)
    return int(best)


class XtPlan:
    """geogcn_xt_plan of one CSR(X^T) structure and output width (include/geogcn.h)."""

    def __init__(self, csr_t: CSR, F: int):
        self._h = C.c_void_p(0)
        lib = _ffi.lib()
        check(lib.geogcn_xt_plan_create(csr_t.shape[0], csr_t.shape[1], csr_t.rowptr_host.ctypes.data_as(C.c_void_p),
                                        csr_t.colidx_host.ctypes.data_as(C.c_void_p), int(F), C.byref(self._h)),
              'xt_plan_create')
        self.F = int(F)
        self.ws_bytes = int(lib.geogcn_xt_workspace_bytes(self._h))

    def __del__(self):
        try:
            if self._h:
                _release(_ffi.lib().geogcn_xt_plan_destroy, self._h)
                self._h = C.c_void_p(0)
        except Exception:
            pass


class _CSRValues:
    """A CSR with the structure (and split plan, workspace) of `base` and its own values -- what value dropout on a
    sparse input produces every step.  Keeps `base` alive; does not own the plan."""

    def __init__(self, base: CSR, val: torch.Tensor):
        self._base = base
        self.val = val
        for k in ('shape', 'nnz', 'rowptr', 'colidx', 'rowptr_host', 'colidx_host', 'device', '_plan', '_ws', 'n_long_rows',
                  'n_chunks'):
            setattr(self, k, getattr(base, k))


def sparse_dropout(x: SparseOperand, p, seed, call):
    """x with every stored value dropped with probability p and the rest scaled by 1/(1-p) (reference
    gcnmodel.py:44-70).  The keep decision is keyed by the element's position, so the forward CSR, the CSR of the
    transposed tail and the dense head panel of the result are still one matrix."""
    lib = _ffi.lib()
    n, V = x.shape
    def csr(c, transposed):
        out = torch.empty_like(c.val)
        check(lib.geogcn_dropout_csr_f32(c.shape[0], _p(c.rowptr), _p(c.colidx), _p(c.val), _p(out), n, V, int(transposed),
                                         float(p), int(seed), int(call), _stream()), 'dropout_csr_f32')
        return _CSRValues(c, out)
    fwd = csr(x.fwd, False)
    if x.bwd is None:
        return SparseOperand(fwd, None, False)
    if x.symmetric or x.bwd is x.fwd:
        # a symmetric matrix stops being symmetric under an element-wise mask: its transpose needs its own values
        bwd = csr(x.fwd, True)
        return SparseOperand(fwd, bwd, False)
    bwd = csr(x.bwd, True)
    head = None
    if x.head_dense is not None:
        head = DMat(x.head_dense.n, x.head_dense.F, x.head_dense.device, ld=x.head_dense.ld)
        check(lib.geogcn_dropout_panel_f32(head.n, head.F, _p(x.head_dense.t), head.ld, _p(x.head_idx), V, float(p), int(seed),
                                           int(call), _p(head.t), _stream()), 'dropout_panel_f32')
    return SparseOperand(fwd, bwd, False, x.head_idx, head)


@_timed('spmm_t')
def spmm_t(x: SparseOperand, G: DMat, out: DMat = None, precision=None):
    """out = x^T . G  (gradient of structured_dot(x, W) w.r.t. W; reference gcnmodel.py:39 autodiff).  `precision`: of the dense
    head panel's product."""
    # column slabs of <= XT_MAX_F: the kernel keeps 2 x F/64 float4 accumulators per lane in registers; up to 320 columns
    # that leaves room for 1024-thread workgroups, beyond it halves the threads and doubles the batches (F = 600 in one
    # piece: 5.5 ms, slower than the row gather's 4.0; as two slabs of 300: see DESIGN.md 4.2)
    n_slab = -(-G.F // tuning.XT_MAX_F)
    width = pad4(-(-G.F // n_slab))
    if x.xt_plan(min(width, G.F)) is not None:
        # tail rows through the document-blocked kernel (every row written; head rows as zeros)
        if out is None:
            out = DMat.empty(x.bwd.shape[0], G.F, G.device)
        for c0 in range(0, G.F, width):
            plan = x.xt_plan(min(width, G.F - c0))
            w = _ws_for(G.device).get(plan.ws_bytes)
            check(_ffi.lib().geogcn_xt_dot_f32(plan._h, _p(x.bwd.colidx), _p(x.bwd.val), C.c_void_p(G.t.data_ptr() + 4 * c0), G.ld,
                                               C.c_void_p(out.t.data_ptr() + 4 * c0), out.ld, _p(w), w.numel(), _stream()),
                  'xt_dot_f32')
    else:
        out = spmm(x.bwd, G, out=out)                 # tail rows (head rows come out as zeros)
    if x.head_dense is not None:
        # K x F on the MFMA pipe, deterministic split-K -- fp32-class in every configuration (the bf16 configuration rounds the
        # H . W products, not the sparse input's gradient)
        p = precision or GEMM_PRECISION              # (resolved HERE: the bf16 configuration may come from GEOGCN_GEMM_PRECISION, precision = None)
        head = gemm(x.head_dense, G, transA=True, precision='bf16x3' if p == 'bf16' else p)
        scatter_rows(head, x.head_idx, out)
    return out


class HotCSR:
    """CSR(X) prepared for geogcn_spmm_csr_hot_f32: the `n_hot` most frequent columns are served from LDS; every row is
    reordered [hot | cold] and a hot entry's column index is its LDS slot (include/geogcn.h)."""

    def __init__(self, csr: CSR, values_host: np.ndarray, n_hot: int):
        indptr, indices = csr.rowptr_host, csr.colidx_host
        counts = np.bincount(indices, minlength=csr.shape[1])
        n_hot = int(min(n_hot, np.count_nonzero(counts)))
        hot = np.sort(np.argsort(-counts, kind='stable')[:n_hot]).astype(np.int32)
        slot_of = np.full(csr.shape[1], -1, dtype=np.int32)
        slot_of[hot] = np.arange(n_hot, dtype=np.int32)
        slot = slot_of[indices]
        is_hot = slot >= 0
        row_of = np.repeat(np.arange(csr.shape[0], dtype=np.int64), np.diff(indptr))
        order = np.lexsort((indices, ~is_hot, row_of))                 # row, hot first, ascending column inside a part
        col2 = np.where(is_hot, slot, indices)[order].astype(np.int32)
        hot_per_row = np.bincount(row_of[is_hot], minlength=csr.shape[0]).astype(np.int32)
        dev = csr.device
        self.shape, self.nnz, self.n_hot = csr.shape, csr.nnz, n_hot
        self.hot_fraction = float(is_hot.mean()) if len(indices) else 0.0
        self.rowptr = csr.rowptr
        self.rowsplit = torch.from_numpy(np.ascontiguousarray(indptr[:-1] + hot_per_row, dtype=np.int32)).to(dev)
        self.colidx = torch.from_numpy(np.ascontiguousarray(col2)).to(dev)
        self.val = torch.from_numpy(np.ascontiguousarray(values_host[order], dtype=np.float32)).to(dev)
        self.hot_rows = torch.from_numpy(hot).to(dev)
        # working order of the rows (speed only): the four rows a wave takes at a time run in lockstep, so rows of similar COLD
        # length (the dependent L2 round trips) go next to each other -- sorted, longest first, then the groups of four dealt round
        # the 256 workgroups so that each gets the same mix and starts with its longest rows
        cold_per_row = (np.diff(indptr) - hot_per_row).astype(np.int64)
        by_len = np.argsort(-cold_per_row, kind='stable').astype(np.int32)
        n = len(by_len)
        n_wg, quad = 256, 4
        groups = -(-n // quad)
        pad = np.full(groups * quad, -1, dtype=np.int32)
        pad[:n] = by_len
        pad = pad.reshape(groups, quad)
        dealt = np.concatenate([pad[c::n_wg] for c in range(n_wg)]).ravel()
        order_rows = dealt[dealt >= 0]
        assert len(order_rows) == n
        self.row_order = torch.from_numpy(np.ascontiguousarray(order_rows, dtype=np.int32)).to(dev) if tuning.HOT_ROW_ORDER else None


def spmm_hot(A: HotCSR, B: DMat, out: DMat = None, bias: torch.Tensor = None, act=ACT_NONE, col0=0, F=None):
    """out = act(A . B + bias), hot rows of B from LDS (geogcn_spmm_csr_hot_f32).  `col0`, `F`: only the column slab
    [col0, col0 + F) of B / bias / out (col0 a multiple of 4) -- a layer wider than the kernel's 384 columns runs as slabs."""
    if B.n != A.shape[1]:
        raise ValueError("spmm_hot: A is %s but B has %d rows" % (A.shape, B.n))
    out = DMat.empty(A.shape[0], B.F, B.device) if out is None else out
    F = B.F - col0 if F is None else int(F)
    if col0 % 4 or col0 < 0 or col0 + F > B.F:
        raise ValueError("spmm_hot: bad column slab [%d, %d) of %d" % (col0, col0 + F, B.F))
    at = lambda t: None if t is None else C.c_void_p(t.data_ptr() + 4 * col0)
    check(_ffi.lib().geogcn_spmm_csr_hot_f32(A.shape[0], B.n, _p(A.rowptr), _p(A.rowsplit), _p(A.colidx), _p(A.val), at(B.t), B.ld,
                                             _p(A.hot_rows), A.n_hot, _p(A.row_order), at(out.t), out.ld, F, at(bias), act, _stream()),
          'spmm_csr_hot_f32')
    return out


@_timed('spmm_x')
def spmm_x(x: SparseOperand, W: DMat, out: DMat = None, bias: torch.Tensor = None, act=ACT_NONE):
    """out = act(x . W + bias) for a sparse input x (S.structured_dot(X, W0), reference gcnmodel.py:39-42): the hot rows
    of W from LDS (geogcn_spmm_csr_hot_f32) from tuning.HOT_MIN_NNZ stored entries on, the plain row gather below.
    (Measured and removed: the dense head panel on the MFMA pipe + CSR tail continuing the rows -- 1.69 ms against
    1.32 -- and its column-slab variant; DESIGN_NOTEBOOK.md section 4.2.)"""
    if isinstance(x.fwd, CSR) and x.fwd.nnz >= tuning.HOT_MIN_NNZ and not x.symmetric:
        # a layer wider than the LDS kernel takes (384 columns) runs as two column slabs of it when they are float4-aligned
        # (600 wide: 2 x 1.33 ms against 3.93 ms for the plain row gather)
        slabs = 1 if int(_ffi.lib().geogcn_spmm_hot_capacity(W.F)) > 0 else (2 if W.F % 8 == 0 else 0)
        cap = int(_ffi.lib().geogcn_spmm_hot_capacity(W.F // slabs)) if slabs else 0
        if cap > 0:
            hot = x._hot.get(cap)
            if hot is None:
                hot = x._hot[cap] = HotCSR(x.fwd, x.fwd.val.cpu().numpy(), cap)
            out = DMat.empty(hot.shape[0], W.F, W.device) if out is None else out
            for s in range(slabs):
                spmm_hot(hot, W, out=out, bias=bias, act=act, col0=s * (W.F // slabs), F=W.F // slabs)
            return out
    return spmm(x.fwd, W, out=out, bias=bias, act=act)


@_timed('spmm_x_dropout')
def spmm_x_dropout(x: SparseOperand, W: DMat, bias, act, p, mask_in=None, seed=0, offset=0, calls_dev=None, per_call=0, base=0):
    """(H0, Hd, mask) = (act(x . W + bias), H0 * keep / (1-p), keep-mask) in ONE launch (geogcn_spmm_csr_hot_dropout_f32):
    the sparse-input layer with the dropout that follows it (reference gcnmodel.py:353,357) in the product's epilogue.  The
    keep decisions are those dropout_mask(n, F, p, seed, offset) / dropout_mask_ctr would draw (bit-identical), or `mask_in`
    (uint8 [n, F] on the device) when a mask is injected.  -> None when the fused kernel does not apply (small X, a width
    that is not a multiple of 4 or beyond the LDS kernel, value-dropped X): the caller then runs the separate kernels."""
    if not (isinstance(x.fwd, CSR) and x.fwd.nnz >= tuning.HOT_MIN_NNZ and not x.symmetric and W.F % 4 == 0
            and act in (ACT_NONE, ACT_TANH) and 0.0 < p < 1.0):
        return None
    lib = _ffi.lib()
    cap = int(lib.geogcn_spmm_hot_capacity(W.F))
    if cap <= 0:
        return None
    hot = x._hot.get(cap)
    if hot is None:
        hot = x._hot[cap] = HotCSR(x.fwd, x.fwd.val.cpu().numpy(), cap)
    n = hot.shape[0]
    if mask_in is not None and not (isinstance(mask_in, torch.Tensor) and mask_in.dtype == torch.uint8 and tuple(mask_in.shape) == (n, W.F)
                                    and mask_in.is_contiguous() and mask_in.device == W.t.device):
        return None              # the kernel reads 32-bit words at mask + row * F + 4 q: anything else goes to the separate kernels
    H0, Hd = DMat.empty(n, W.F, W.device), DMat.empty(n, W.F, W.device)
    mask = mask_in if mask_in is not None else torch.empty((n, W.F), dtype=torch.uint8, device=W.device)
    check(lib.geogcn_spmm_csr_hot_dropout_f32(n, W.n, _p(hot.rowptr), _p(hot.rowsplit), _p(hot.colidx), _p(hot.val), _p(W.t), W.ld,
                                              _p(hot.hot_rows), hot.n_hot, _p(hot.row_order), _p(H0.t), _p(Hd.t), H0.ld, W.F, _p(bias), int(act),
                                              float(p), _p(mask_in), None if mask_in is not None else _p(mask), int(seed),
                                              int(offset), _p(calls_dev), int(per_call), int(base), _stream()),
          'spmm_csr_hot_dropout_f32')
    return H0, Hd, mask


class SpmmTimer:
    """hipEvent pairs around the SpMM row kernel, recorded by the library on the launch stream
    (bench.py's roofline leg; torch.cuda.Event would also do, the library pool avoids per-call
    Python work inside the timed region)."""

    def __init__(self, capacity=4096):
        self._h = C.c_void_p(0)
        check(_ffi.lib().geogcn_timer_create(int(capacity), C.byref(self._h)), 'timer_create')
        self.capacity = int(capacity)

    def attach(self, csr: CSR, only_F=0):
        """Sample the plain products that run on `csr`'s plan (width only_F; 0 = any).  The timer is a handle held by the
        caller and attached to that one plan: the library keeps no global state."""
        self.detach()
        check(_ffi.lib().geogcn_spmm_plan_attach_timer(csr._plan, self._h, int(only_F)), 'spmm_plan_attach_timer')
        self._csr = csr

    def detach(self):
        csr = getattr(self, '_csr', None)
        if csr is not None and csr._plan:
            check(_ffi.lib().geogcn_spmm_plan_attach_timer(csr._plan, None, 0), 'spmm_plan_attach_timer')
        self._csr = None

    def read_ms(self):
        buf = (C.c_float * self.capacity)()
        n = C.c_int32(0)
        check(_ffi.lib().geogcn_timer_read_ms(self._h, buf, self.capacity, C.byref(n)), 'timer_read_ms')
        return [buf[i] for i in range(n.value)]

    def __del__(self):
        try:
            self.detach()
            if self._h:
                _ffi.lib().geogcn_timer_destroy(self._h)
                self._h = C.c_void_p(0)
        except Exception:
            pass
