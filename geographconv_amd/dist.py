"""1-D row partition of the full-graph GCN across the GPUs of one node.

The reference has no distributed code at all (single process, gcnmodel.py:409-430).  Here rank r
owns the contiguous row block [r*R, min(N, (r+1)*R)) of X, A_hat, H, Y with R = ceil(N / world);
column indices of A_hat stay global.  Per graph-convolution layer and direction there is ONE
exchange: the locally produced rows of Z = H.W (forward) or dS (backward) are all-gathered --
in place, each rank's GEMM/elementwise kernel having written straight into its slot of the
gathered buffer -- and the local SpMM then reads the whole gathered matrix.  Parameter gradients
and the four loss/accuracy sums are all-reduced once per step over a flat arena.  Collectives go
through ``torch.distributed`` (backend "nccl" = RCCL over xGMI on the GPUs, "gloo" in CPU tests)."""
from __future__ import annotations

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

    def split_indices(self, idx: np.ndarray, y: np.ndarray = None):
        """Global row indices -> (local indices, selected labels) for the rows this rank owns."""
        idx = np.asarray(idx)
        sel = (idx >= self.r0) & (idx < self.r1)
        loc = (idx[sel] - self.r0).astype(np.int32)
        return (loc, None if y is None else np.asarray(y)[sel].astype(np.int32), sel)


class Comm:
    """Single-rank communicator: every exchange is the identity."""
    rank, world = 0, 1

    def __init__(self, N=None, device=None):
        self.part = None if N is None else RowPartition(N, 1, 0)
        self.device = device

    def gather_buffer(self, F, tag=None):
        """(gathered matrix, view of the local slot).  Single rank: one buffer, no pad rows."""
        K = backend.active()
        n = self.part.N
        buf = K.DMat.empty(n, F, self.device, ld=K.gather_ld(F))
        return buf, buf

    def all_gather_rows_(self, gathered):
        return gathered

    def all_reduce_sum_(self, t: torch.Tensor):
        return t


class TorchDistComm(Comm):
    """One process per GPU; RCCL (or gloo) through torch.distributed."""

    def __init__(self, N, device, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.part = RowPartition(N, self.world, self.rank)
        self.device = device
        self._bufs = {}

    def gather_buffer(self, F, tag=None):
        K = backend.active()
        key = (int(F), tag)
        buf = self._bufs.get(key)
        if buf is None:
            # zero once: the tail rows of the last slot are never written and must read as 0
            buf = self._bufs[key] = K.DMat(self.part.n_gathered, F, self.device, ld=K.gather_ld(F))
        lo = self.rank * self.part.R
        return buf, buf.rows(lo, lo + self.part.n_local)

    def all_gather_rows_(self, gathered):
        """In-place all-gather: every rank contributes its R-row slot of `gathered`."""
        R = self.part.R
        slot = gathered.t[self.rank * R:(self.rank + 1) * R]
        self.dist.all_gather_into_tensor(gathered.t, slot, group=self.group)
        return gathered

    def all_reduce_sum_(self, t: torch.Tensor):
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return t
