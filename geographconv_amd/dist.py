"""Full-graph GCN across the GPUs of one node.  The reference has no distributed
code at all (single process, gcnmodel.py:409-430).

Rank r owns the contiguous row block [r*R, min(N, (r+1)*R)) of X, H, Y (R = ceil(N / world));
parameters are replicated; parameter gradients and the four loss/accuracy sums are all-reduced
once per step over a flat arena.  The only per-layer exchange is around the graph convolution
S = A_hat . Z (forward) / dZ = A_hat^T . dS (backward), with two interchangeable schemes:

``a2a`` (default from 3 ranks) -- REPARTITION BY FEATURES.  Every rank keeps the whole (88 MB) A_hat.  The
    row-partitioned Z (n_local x F) is packed into `world` feature panels of width wp = ceil4(F/world)
    and exchanged with ONE all-to-all, so that rank q holds panel q of ALL rows (N x wp); it runs
    the SpMM on that narrow operand (bias + activation fused, they are per column) and a second
    all-to-all returns the result to the row partition.  Per rank and exchange 2*(w-1)/w * N/w * F
    floats move -- 116 MB at w = 8, F = 300 -- instead of the (w-1)/w * N * F (462 MB) an all-gather
    would deliver to every rank, and an all-to-all keeps all 7 xGMI links of a GPU busy at once.
``allgather`` -- 1-D row split of A_hat as well: the local GEMM writes Z_r straight into rank r's
    slot of the gathered buffer, all_gather_into_tensor runs in place, the local SpMM reads the
    whole gathered matrix.  Simple, but communication-bound beyond 2 GPUs on a graph without locality.

Collectives go through ``torch.distributed`` (backend "nccl" = RCCL over xGMI on the GPUs, "gloo" in
the CPU tests).  GEOGCN_DIST_EXCHANGE=a2a|allgather overrides the default (all-gather at 2 ranks, a2a from 3)."""
from __future__ import annotations

import os

import numpy as np
import scipy.sparse as sps
import torch

from . import backend


class RowPartition:
    def __init__(self, N: int, world: int, rank: int):
        self.N, self.world, self.rank = int(N), int(world), int(rank)
        self.R = (self.N + self.world - 1) // self.world          # rows per slot
        self.r0 = min(self.N, self.rank * self.R)
        self.r1 = min(self.N, self.r0 + self.R)
        self.n_local = self.r1 - self.r0
        self.n_gathered = self.R * self.world                       # >= N; tail rows stay zero

    def bounds(self, rank):
        r0 = min(self.N, rank * self.R)
        return r0, min(self.N, r0 + self.R)

    def local_rows(self, m):
        """Row block of a scipy matrix / numpy array."""
        return m[self.r0:self.r1]

    def local_rows_csr(self, m: sps.spmatrix, pad_cols_to=None):
        """Local row block of a CSR matrix, columns widened to the gathered height so that the
        SpMM's n_cols matches the gathered operand (extra columns are empty)."""
        blk = sps.csr_matrix(m)[self.r0:self.r1]
        if pad_cols_to is not None and pad_cols_to != blk.shape[1]:
            blk = sps.csr_matrix((blk.data, blk.indices, blk.indptr), shape=(blk.shape[0], pad_cols_to))
        return blk

    def padded_square_csr(self, m: sps.spmatrix):
        """The whole matrix grown to n_gathered x n_gathered (empty tail rows / columns): the operand
        of the feature-partitioned SpMM, whose dense operand is laid out by gathered row index."""
        m = sps.csr_matrix(m)
        g = self.n_gathered
        if m.shape == (g, g):
            return m
        indptr = np.concatenate([m.indptr, np.full(g - m.shape[0], m.indptr[-1], dtype=m.indptr.dtype)])
        return sps.csr_matrix((m.data, m.indices, indptr), shape=(g, g))

    def split_indices(self, idx: np.ndarray, y: np.ndarray = None):
        """Global row indices -> (local indices, selected labels) for the rows this rank owns."""
        idx = np.asarray(idx)
        sel = (idx >= self.r0) & (idx < self.r1)
        loc = (idx[sel] - self.r0).astype(np.int32)
        return (loc, None if y is None else np.asarray(y)[sel].astype(np.int32), sel)


class Comm:
    """Single-rank communicator: every exchange is the identity."""
    rank, world = 0, 1
    exchange = 'none'

    def __init__(self, N=None, device=None):
        self.part = None if N is None else RowPartition(N, 1, 0)
        self.device = device

    def graph_operand(self, A_host, hub_row_bytes=None):
        K = backend.active()
        return K.SparseOperand.from_scipy(A_host, self.device, dense_head=False, hub_row_bytes=hub_row_bytes)

    def matmul_target(self, F, tag=None):
        """Where the GEMM that produces the SpMM's dense operand should write (n_local x F)."""
        K = backend.active()
        return K.DMat.empty(self.part.N, F, self.device, ld=K.gather_ld(F))

    def stage_operand(self, m, F, tag=None):
        """An existing matrix as the SpMM's dense operand (the backward's dS): used in place on one GPU; the
        partitioned communicator copies it into its exchange buffer."""
        return m

    def graph_spmm(self, A_csr, z, bias, act, F, tag=None):
        return self.graph_spmm_end(self.graph_spmm_begin(A_csr, z, bias, act, F, tag))

    # two-phase form: `begin` starts the exchange of Z, `end` waits for it and multiplies.  Work the caller
    # enqueues between the two (the highway gate's GEMMs) overlaps the collective.
    def graph_spmm_begin(self, A_csr, z, bias, act, F, tag=None):
        return dict(A=A_csr, z=z, bias=bias, act=act, F=F, tag=tag, work=None)

    def graph_spmm_end(self, h):
        return backend.active().spmm(h['A'], h['z'], bias=h['bias'], act=h['act'], F=h['F'])

    def all_gather_rows(self, local):
        return local

    def all_reduce_sum_(self, t: torch.Tensor):
        return t


class StreamOverlapComm(Comm):
    """Single GPU: the graph SpMM runs on a SIDE HIP stream between graph_spmm_begin and _end, so that the work the
    layer sweep enqueues in between (the highway gate's GEMMs) shares the device with it.  The SpMM is bound by
    gather traffic from beyond the L2 and leaves MFMA pipes and most issue slots idle; the fp32 GEMM is
    MFMA-bound -- measured with tools/cu_mask_probe.py: (SpMM, gate GEMM) 2.70 -> 2.50 ms, (SpMM^T, dWt, dH)
    3.59 -> 3.34 ms.  Same kernels, same arithmetic; only the launch streams differ."""
    exchange = 'stream'

    def __init__(self, N, device):
        super().__init__(N, device)
        self.side = torch.cuda.Stream(device=device)

    def graph_spmm_begin(self, A_csr, z, bias, act, F, tag=None):
        K = backend.active()
        main = torch.cuda.current_stream(self.device)
        ready = torch.cuda.Event()
        ready.record(main)
        self.side.wait_event(ready)                  # the operand was produced on the main stream
        with torch.cuda.stream(self.side):
            out = K.spmm(A_csr, z, bias=bias, act=act, F=F)
            done = torch.cuda.Event()
            done.record(self.side)
        # caching-allocator bookkeeping: both tensors are used on a stream other than the one that allocated them
        z.t.record_stream(self.side)
        out.t.record_stream(main)
        if bias is not None:
            bias.record_stream(self.side)
        return dict(out=out, done=done, work=None)

    def graph_spmm_end(self, h):
        torch.cuda.current_stream(self.device).wait_event(h['done'])
        return h['out']


class TorchDistComm(Comm):
    """One process per GPU; RCCL (or gloo) through torch.distributed."""

    def __init__(self, N, device, group=None, exchange=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.part = RowPartition(N, self.world, self.rank)
        self.device = device
        self.exchange = exchange or os.environ.get('GEOGCN_DIST_EXCHANGE', 'auto')
        if self.exchange == 'auto':
            # at 2 ranks both schemes move the same bytes and the all-gather needs no repacking; from 3 ranks on
            # the all-to-all moves (w-1)/w * 2/w of what the all-gather delivers to every rank
            self.exchange = 'a2a' if self.world >= 3 else 'allgather'
        if self.exchange not in ('a2a', 'allgather'):
            raise ValueError("GEOGCN_DIST_EXCHANGE must be 'a2a' or 'allgather', got %r" % self.exchange)
        self._bufs = {}

    # -- constant operand ----------------------------------------------------------------------------
    def graph_operand(self, A_host, hub_row_bytes=None):
        """A_hat as this exchange scheme needs it: the whole matrix (a2a) or the local row block with
        the column space widened to the gathered height (allgather)."""
        K = backend.active()
        part = self.part
        A_csr = sps.csr_matrix(A_host).astype(np.float32)
        At = sps.csr_matrix(A_csr.T)
        if self.exchange == 'a2a':
            Af, Ab = part.padded_square_csr(A_csr), part.padded_square_csr(At)
        else:
            Af = part.local_rows_csr(A_csr, part.n_gathered)
            Ab = part.local_rows_csr(At, part.n_gathered)
        Af.sort_indices()
        Ab.sort_indices()
        same = (Af.shape == Ab.shape and np.array_equal(Af.indptr, Ab.indptr) and np.array_equal(Af.indices, Ab.indices)
                and np.array_equal(Af.data, Ab.data))
        fwd = K.CSR(Af, self.device, hub_row_bytes=hub_row_bytes)
        bwd = fwd if same else K.CSR(Ab, self.device, hub_row_bytes=hub_row_bytes)
        return K.SparseOperand(fwd, bwd, same)

    # -- buffers ---------------------------------------------------------------------------------------
    def _gather_buffer(self, F, tag):
        K = backend.active()
        key = ('ag', int(F), tag)
        buf = self._bufs.get(key)
        if buf is None:
            # zero once: the tail rows of the last slot are never written and must read as 0
            buf = self._bufs[key] = K.DMat(self.part.n_gathered, F, self.device, ld=K.gather_ld(F))
        return buf

    def _flat(self, key, numel):
        t = self._bufs.get(key)
        if t is None or t.numel() != numel:
            t = self._bufs[key] = torch.zeros(numel, dtype=torch.float32, device=self.device)
        return t

    def matmul_target(self, F, tag=None):
        K = backend.active()
        if self.exchange == 'allgather':
            buf = self._gather_buffer(F, tag)
            lo = self.rank * self.part.R
            return buf.rows(lo, lo + self.part.n_local)         # the GEMM writes straight into my slot
        return K.DMat.empty(self.part.n_local, F, self.device)

    def stage_operand(self, m, F, tag=None):
        g = self.matmul_target(F, tag=tag)
        g.copy_from(m)
        return g

    # -- the exchange around A . Z ----------------------------------------------------------------------
    def panel_width(self, F):
        K = backend.active()
        return K.pad4((int(F) + self.world - 1) // self.world)

    def graph_spmm_begin(self, A_csr, z, bias, act, F, tag=None):
        """Start act(A . Z + bias) for row-partitioned Z: the exchange of Z is issued asynchronously (RCCL runs it
        on its own stream); finish with graph_spmm_end."""
        K = backend.active()
        part, W = self.part, self.world
        h = dict(A=A_csr, bias=bias, act=act, F=F, tag=tag)
        if self.exchange == 'allgather':
            buf = self._gather_buffer(F, tag)
            R = part.R
            h['buf'] = buf
            h['work'] = self.dist.all_gather_into_tensor(buf.t, buf.t[self.rank * R:(self.rank + 1) * R], group=self.group,
                                                         async_op=True)
            return h
        R, wp = part.R, self.panel_width(F)
        n = W * R * wp
        send = self._flat(('send', wp, tag), n)
        recv = self._flat(('recv', wp, tag), n)
        K.pack_panels(z, R, W, wp, send)
        h.update(send=send, recv=recv, wp=wp, n=n)
        h['work'] = self.dist.all_to_all_single(recv, send, group=self.group, async_op=True)
        return h

    def graph_spmm_end(self, h):
        """-> row-partitioned result (n_local x F)."""
        K = backend.active()
        part, W = self.part, self.world
        if h['work'] is not None:
            h['work'].wait()            # nccl: the compute stream waits for the collective; gloo: the host does
        A_csr, bias, act, F, tag = h['A'], h['bias'], h['act'], h['F'], h['tag']
        if self.exchange == 'allgather':
            return K.spmm(A_csr, h['buf'], bias=bias, act=act, F=F)
        R, wp, n, send, recv = part.R, h['wp'], h['n'], h['send'], h['recv']
        zslab = K.DMat(part.n_gathered, wp, t=recv.view(part.n_gathered, wp))     # panel `rank` of ALL rows
        bslab = None
        if bias is not None:
            bp = torch.zeros(W * wp, dtype=torch.float32, device=self.device)
            bp[:F].copy_(bias[:F])
            bslab = bp[self.rank * wp:(self.rank + 1) * wp]
        sslab = K.DMat(part.n_gathered, wp, t=send.view(part.n_gathered, wp))      # reuse the send buffer
        K.spmm(A_csr, zslab, out=sslab, bias=bslab, act=act, F=wp)
        back = self._flat(('back', wp, tag), n)
        self.dist.all_to_all_single(back, send, group=self.group)
        out = K.DMat.empty(part.n_local, F, self.device)
        K.unpack_panels(back, R, W, wp, out)
        return out

    def all_gather_rows(self, local):
        """(n_local x F) -> (N x F) on every rank (used only to hand the full probability matrix back)."""
        K = backend.active()
        buf = self._gather_buffer(local.F, 'out')
        lo = self.rank * self.part.R
        buf.rows(lo, lo + self.part.n_local).copy_from(local)
        R = self.part.R
        self.dist.all_gather_into_tensor(buf.t, buf.t[self.rank * R:(self.rank + 1) * R], group=self.group)
        return buf

    def all_reduce_sum_(self, t: torch.Tensor):
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return t
