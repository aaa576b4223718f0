"""Full-graph GCN across the GPUs of one node.  The reference has no distributed
code at all (single process, gcnmodel.py:409-430).

Rank r owns a contiguous row block of X, H, Y; parameters are replicated; parameter gradients and the four
loss/accuracy sums are all-reduced once per step over a flat arena.  The only per-layer exchange is around the graph
convolution S = A_hat . Z (forward) / dZ = A_hat^T . dS (backward), with two interchangeable schemes:

``a2a`` (default from 3 ranks) -- REPARTITION BY FEATURES.  Every rank keeps the whole (88 MB) A_hat.  The
    row-partitioned Z (n_local x F) is laid out as `world` feature panels of width wp = ceil4(F/world) -- the GEMM that
    produces it writes that layout straight from its accumulators (ops.Panels, geogcn_gemm_panels_f32) -- and exchanged
    with ONE all-to-all, so that rank q holds panel q of ALL rows (N x wp); it runs the SpMM on that narrow operand
    (bias + activation fused, they are per column) and a second all-to-all returns the result to the row
    partition.  Per rank and exchange 2*(w-1)/w * N/w * F floats move -- 116 MB at w = 8, F = 300 -- instead of the
    (w-1)/w * N * F (462 MB) an all-gather would deliver to every rank, and an all-to-all keeps all 7 xGMI links of a GPU
    busy at once.  Both all-to-alls are asynchronous: the caller runs independent work between begin / mid / end.
``allgather`` -- 1-D row split of A_hat as well (the north_star's scheme): the local GEMM writes Z_r straight into rank
    r's slot of the gathered buffer, all_gather_into_tensor runs in place, the local SpMM reads the whole gathered
    matrix.  The row split is balanced by COST (stored edges + a per-row term), not by row count: on a power-law graph a
    rank holding the hubs would otherwise do several times the SpMM work of the others.

``halo`` -- the all-gather scheme for graphs WITH locality (communities numbered contiguously, `GraphConv(reorder=...)`): the
    same cost-balanced 1-D row split, but a rank receives only the rows of Z its block of A_hat actually references (its
    HALO) instead of all of Z: the owner packs them (one row-gather launch) and ONE all-to-all with per-peer sizes delivers
    them behind the rank's own rows, the local SpMM reads [own rows | halo] through renumbered columns (stored order kept:
    bitwise the one-GPU accumulation).  On the pinned power-law graph a block of 55,000 rows references 95 % of all nodes
    -- no better than the all-gather; on a community graph the halo is the endpoints of the ~11 % global edges.  `auto`
    picks it when the largest halo is at most tuning.DIST_HALO_MAX_FRACTION (0.9 at two ranks) of the remote rows.

In the bf16 configuration (`gemm_precision='bf16'`, BASELINE config 5) the exchanged operand is bfloat16 in both schemes
(the GEMM stores it as such; dS is cast while it is staged): half the bytes on the wire and per gathered row.

Collectives go through ``torch.distributed`` (backend "nccl" = RCCL over xGMI on the GPUs, "gloo" in
the CPU tests).  GEOGCN_DIST_EXCHANGE=a2a|allgather|halo overrides the default (halo where the graph allows it, else
all-gather at 2 ranks, a2a from 3)."""
from __future__ import annotations

import ctypes as C

import numpy as np
import scipy.sparse as sps
import torch

from . import backend, tuning

# cost of one row in stored-edge equivalents when balancing the all-gather scheme's row split (tuning.py)
ROW_COST_IN_EDGES = tuning.ROW_COST_IN_EDGES


def balanced_bounds(indptr, world, row_cost=ROW_COST_IN_EDGES):
    """Row boundaries b[0..world] (b[0] = 0, b[world] = N) that equalise sum(nnz(row) + row_cost) per block."""
    indptr = np.asarray(indptr, dtype=np.int64)
    N = len(indptr) - 1
    cum = indptr + row_cost * np.arange(N + 1, dtype=np.float64)          # cost of rows [0, i)
    targets = cum[-1] * np.arange(1, world, dtype=np.float64) / world
    cuts = np.searchsorted(cum, targets, side='left')
    return np.concatenate([[0], np.minimum(cuts, N), [N]]).astype(np.int64)


class RowPartition:
    """Contiguous row blocks [bounds[r], bounds[r+1]).  Exchange buffers give every rank a SLOT of R = max block size
    rows: global row g of rank r sits at slot position r * R + (g - bounds[r]).  With the default uniform split
    (R = ceil(N / world)) slot position == global index."""

    def __init__(self, N: int, world: int, rank: int, bounds=None):
        self.N, self.world, self.rank = int(N), int(world), int(rank)
        if bounds is None:
            R = (self.N + self.world - 1) // self.world
            bounds = np.minimum(self.N, R * np.arange(self.world + 1, dtype=np.int64))
            self.uniform = True
        else:
            bounds = np.asarray(bounds, dtype=np.int64)
            assert len(bounds) == self.world + 1 and bounds[0] == 0 and bounds[-1] == self.N and np.all(np.diff(bounds) >= 0)
            R = int(np.diff(bounds).max()) if self.world else 0
            self.uniform = False
        self.bounds_all = bounds
        self.R = max(int(R), 1) if self.N else int(R)
        self.r0, self.r1 = int(bounds[self.rank]), int(bounds[self.rank + 1])
        self.n_local = self.r1 - self.r0
        self.n_gathered = self.R * self.world                       # >= N; unused slot rows stay zero

    def bounds(self, rank):
        return int(self.bounds_all[rank]), int(self.bounds_all[rank + 1])

    def slot_position(self, g):
        """Global row indices -> positions in the gathered (slot) layout."""
        g = np.asarray(g, dtype=np.int64)
        owner = np.searchsorted(self.bounds_all, g, side='right') - 1
        owner = np.clip(owner, 0, self.world - 1)
        return owner * self.R + (g - self.bounds_all[owner])

    def local_rows(self, m):
        """Row block of a scipy matrix / numpy array."""
        return m[self.r0:self.r1]

    def _remap_columns(self, m: sps.csr_matrix, n_cols):
        """Same stored order (= same accumulation order as on one GPU), column ids replaced by slot positions."""
        cols = self.slot_position(m.indices).astype(np.int32) if not self.uniform else m.indices
        return sps.csr_matrix((m.data, cols, m.indptr), shape=(m.shape[0], n_cols))

    def local_rows_csr(self, m: sps.spmatrix, pad_cols_to=None):
        """Local row block of a CSR matrix whose columns index the gathered operand (slot positions; the extra
        columns are empty)."""
        blk = sps.csr_matrix(sps.csr_matrix(m)[self.r0:self.r1])
        width = blk.shape[1] if pad_cols_to is None else pad_cols_to
        if self.uniform and width == blk.shape[1]:
            return blk
        return self._remap_columns(blk, width)

    def padded_square_csr(self, m: sps.spmatrix):
        """The whole matrix in slot coordinates, n_gathered x n_gathered (empty rows / columns where a slot is not
        full): the operand of the feature-partitioned SpMM, whose dense operand and result are laid out by slot."""
        m = sps.csr_matrix(m)
        g = self.n_gathered
        if m.shape == (g, g):
            return m
        if self.uniform:
            indptr = np.concatenate([m.indptr, np.full(g - m.shape[0], m.indptr[-1], dtype=m.indptr.dtype)])
            return sps.csr_matrix((m.data, m.indices, indptr), shape=(g, g))
        counts = np.zeros(g, dtype=np.int64)
        counts[self.slot_position(np.arange(m.shape[0]))] = np.diff(m.indptr)
        indptr = np.concatenate([[0], np.cumsum(counts)]).astype(m.indptr.dtype)
        # rows keep their relative order inside each block and blocks are ascending: the data order is unchanged
        return sps.csr_matrix((m.data, self.slot_position(m.indices).astype(np.int32), indptr), shape=(g, g))

    def split_indices(self, idx: np.ndarray, y: np.ndarray = None):
        """Global row indices -> (local indices, selected labels) for the rows this rank owns."""
        idx = np.asarray(idx)
        sel = (idx >= self.r0) & (idx < self.r1)
        loc = (idx[sel] - self.r0).astype(np.int32)
        return (loc, None if y is None else np.asarray(y)[sel].astype(np.int32), sel)


def _pattern_rows(A_csr, r0, r1, symmetric):
    """Rows [r0, r1) of the pattern of A + A^T as (indptr, indices) -- the halo of a row block must serve the forward
    operand (A's block) and the backward one (A^T's block) alike; for a symmetric A that is A's own block."""
    blk = A_csr[r0:r1]
    if symmetric:
        return blk.indptr.astype(np.int64), blk.indices.astype(np.int64)
    both = sps.csr_matrix(blk != 0) + sps.csr_matrix(sps.csr_matrix(A_csr.T)[r0:r1] != 0)
    both.sort_indices()
    return both.indptr.astype(np.int64), both.indices.astype(np.int64)


def halo_sizes(A_csr, bounds, symmetric):
    """Rows every rank would RECEIVE per exchange under the halo scheme: distinct columns of its block (of A + A^T) that
    other ranks own.  One pass over the whole matrix, the same on every rank (the `auto` choice must be global)."""
    A_csr = sps.csr_matrix(A_csr)
    N, world = A_csr.shape[0], len(bounds) - 1
    if not symmetric:
        A_csr = sps.csr_matrix((A_csr != 0) + (sps.csr_matrix(A_csr.T) != 0))
    rows = np.repeat(np.arange(N, dtype=np.int64), np.diff(A_csr.indptr))
    cols = A_csr.indices.astype(np.int64)
    own_r = np.searchsorted(bounds, rows, side='right') - 1
    own_c = np.searchsorted(bounds, cols, side='right') - 1
    far = own_r != own_c
    pairs = np.unique(own_r[far] * N + cols[far])
    return np.bincount(pairs // N, minlength=world).astype(np.int64)


class HaloPlan:
    """Who sends which rows to whom under the halo scheme, computed by every rank from ITS OWN row block of the pattern of
    A + A^T (symmetric as a pattern, so rank q's receive list from me -- the columns in my range its rows reference, in
    ascending order -- is my send list to q: the rows of mine that have an entry in q's range, ascending.  No negotiation).

    Operand layout on rank r: rows [0, n_local) = its own block, then the halo: the rows received from rank 0, 1, ... (each
    piece ascending by global index, i.e. the whole halo ascending).  `position(cols)` renumbers global column ids."""

    def __init__(self, part: RowPartition, indptr, indices):
        self.part = part
        r0, r1, world, rank = part.r0, part.r1, part.world, part.rank
        bounds = part.bounds_all
        indices = np.asarray(indices, dtype=np.int64)
        remote = np.unique(indices[(indices < r0) | (indices >= r1)])
        self.recv_cols = remote                                                   # global ids, ascending = grouped by owner
        owner = np.searchsorted(bounds, remote, side='right') - 1
        self.recv_counts = np.bincount(owner, minlength=world).astype(np.int64)
        rows = np.repeat(np.arange(part.n_local, dtype=np.int64), np.diff(indptr))
        own_c = np.searchsorted(bounds, indices, side='right') - 1
        far = own_c != rank
        key = np.unique(own_c[far] * max(part.n_local, 1) + rows[far])            # (destination, local row), ascending
        self.send_rows = (key % max(part.n_local, 1)).astype(np.int32)
        self.send_counts = np.bincount(key // max(part.n_local, 1), minlength=world).astype(np.int64)
        self.n_halo, self.n_send = int(len(remote)), int(len(key))
        self.n_operand = part.n_local + self.n_halo

    def position(self, cols):
        cols = np.asarray(cols, dtype=np.int64)
        own = (cols >= self.part.r0) & (cols < self.part.r1)
        at = np.searchsorted(self.recv_cols, cols)
        found = self.recv_cols[np.minimum(at, self.n_halo - 1)] == cols if self.n_halo else np.zeros(len(cols), dtype=bool)
        if not np.all(own | found):
            raise ValueError("halo exchange: the matrix references rows outside the halo of the adjacency the plan was "
                             "built from (TorchDistComm.prepare must see the full graph first)")
        return np.where(own, cols - self.part.r0, self.part.n_local + at).astype(np.int32)

    def local_rows_csr(self, m):
        """Row block of a CSR matrix with its columns renumbered to operand positions, STORED ORDER KEPT (own columns are
        no longer ascending next to halo ones; the SpMM accumulates in stored order -- the one-GPU order)."""
        blk = sps.csr_matrix(sps.csr_matrix(m)[self.part.r0:self.part.r1])
        return sps.csr_matrix((blk.data, self.position(blk.indices), blk.indptr), shape=(blk.shape[0], max(self.n_operand, 1)))


class Comm:
    """Single-rank communicator: every exchange is the identity."""
    rank, world = 0, 1
    exchange = 'none'

    def __init__(self, N=None, device=None):
        self.part = None if N is None else RowPartition(N, 1, 0)
        self.device = device

    @property
    def capturable(self):
        """May a whole partitioned f_train step be recorded into a hipGraph and replayed (GraphConv(hip_graph=True))?  Only when every
        collective is a stream-ordered device operation: RCCL through torch.distributed's nccl backend or the library's geogcn_comm_*
        entry points (RCCL 2.26 collectives are capturable).  Host-staged transports (gloo, staged-gloo) are not."""
        return False

    def prepare(self, A_host):
        """Hook called with the host adjacency before anything is partitioned (row-split balancing)."""

    def graph_operand(self, A_host):
        K = backend.active()
        return K.SparseOperand.from_scipy(A_host, self.device, dense_head=False)

    def matmul_target(self, F, tag=None, precision=None, direct=True):
        """Where the GEMM that produces the SpMM's dense operand should write (n_local x F)."""
        K = backend.active()
        return K.DMat.empty(self.part.N, F, self.device, ld=K.gather_ld(F))

    def stage_operand(self, m, F, tag=None, precision=None):
        """An existing matrix as the SpMM's dense operand (the backward's dS): used in place on one GPU; the
        partitioned communicator copies it into its exchange buffer."""
        return m

    def graph_spmm(self, A_csr, z, bias, act, F, tag=None):
        return self.graph_spmm_end(self.graph_spmm_begin(A_csr, z, bias, act, F, tag))

    # three-phase form: `begin` starts the exchange of Z, `mid` waits for it, multiplies and starts the return exchange,
    # `end` waits for that.  Work the caller enqueues between the phases overlaps the collectives.
    def graph_spmm_begin(self, A_csr, z, bias, act, F, tag=None):
        return dict(A=A_csr, z=z, bias=bias, act=act, F=F, tag=tag, work=None)

    def graph_spmm_mid(self, h):
        return h

    def graph_spmm_end(self, h):
        return backend.active().spmm(h['A'], h['z'], bias=h['bias'], act=h['act'], F=h['F'])

    def all_gather_rows(self, local):
        return local

    def all_reduce_sum_(self, t: torch.Tensor):
        return t


class _StreamWork:
    """What an asynchronous NativeRccl collective returns: wait() orders the CURRENT stream after it (no host block)."""

    def __init__(self, done):
        self._done = done

    def wait(self):
        torch.cuda.current_stream().wait_event(self._done)
        return True


class NativeRccl:
    """The three collectives TorchDistComm uses, through the library's OWN RCCL entry points (include/geogcn.h
    geogcn_comm_*) instead of torch.distributed -- same call shapes, so TorchDistComm does not care which it holds
    (`GEOGCN_DIST_BACKEND=native`).  A synchronous call is enqueued on the current stream; an asynchronous one on a side
    stream that first waits for the current stream (the semantics of torch's NCCL process group)."""

    class ReduceOp:
        SUM = 'sum'

    def __init__(self, world, rank, unique_id: bytes, device):
        from . import _ffi
        self._ffi = _ffi
        self._lib = _ffi.lib()
        if not self._lib.geogcn_comm_available():
            raise RuntimeError("GEOGCN_DIST_BACKEND=native: RCCL is not available in this process")
        torch.cuda.set_device(device)
        self._h = C.c_void_p(0)
        idbuf = C.create_string_buffer(bytes(unique_id), _ffi.COMM_ID_BYTES)
        _ffi.check(self._lib.geogcn_comm_init_rank(idbuf, int(world), int(rank), C.byref(self._h)), 'comm_init_rank')
        self.world, self.rank = int(world), int(rank)
        self.side = torch.cuda.Stream(device=device)

    @staticmethod
    def unique_id() -> bytes:
        from . import _ffi
        buf = C.create_string_buffer(_ffi.COMM_ID_BYTES)
        _ffi.check(_ffi.lib().geogcn_comm_unique_id(buf, _ffi.COMM_ID_BYTES), 'comm_unique_id')
        return buf.raw

    def close(self):
        if getattr(self, '_h', None):
            self._lib.geogcn_comm_destroy(self._h)
            self._h = C.c_void_p(0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _run(self, enqueue, async_op, tensors):
        cur = torch.cuda.current_stream()
        if not async_op:
            enqueue(C.c_void_p(cur.cuda_stream))
            return None
        ready = torch.cuda.Event()
        ready.record(cur)
        self.side.wait_event(ready)
        enqueue(C.c_void_p(self.side.cuda_stream))
        done = torch.cuda.Event()
        done.record(self.side)
        for t in tensors:
            t.record_stream(self.side)
        return _StreamWork(done)

    def all_reduce(self, t, op=None, group=None, async_op=False):
        assert t.dtype == torch.float32 and t.is_contiguous()
        return self._run(lambda st: self._ffi.check(self._lib.geogcn_comm_allreduce_sum_f32(
            self._h, C.c_void_p(t.data_ptr()), t.numel(), st), 'comm_allreduce_sum_f32'), async_op, (t,))

    def all_gather_into_tensor(self, out, inp, group=None, async_op=False):
        nbytes = inp.numel() * inp.element_size()
        assert out.is_contiguous() and inp.is_contiguous() and out.numel() * out.element_size() == nbytes * self.world
        return self._run(lambda st: self._ffi.check(self._lib.geogcn_comm_allgather(
            self._h, C.c_void_p(inp.data_ptr()), C.c_void_p(out.data_ptr()), nbytes, st), 'comm_allgather'), async_op, (out, inp))

    def all_to_all_single(self, out, inp, output_split_sizes=None, input_split_sizes=None, group=None, async_op=False):
        if output_split_sizes is not None:
            # the halo exchange: per-peer ROW counts of 2-D buffers with one pitch (geogcn_comm_alltoallv takes bytes)
            assert out.is_contiguous() and inp.is_contiguous() and out.dim() == 2 and inp.dim() == 2
            rb = inp.shape[1] * inp.element_size()
            assert rb == out.shape[1] * out.element_size()
            sb = (C.c_int64 * self.world)(*[int(s) * rb for s in input_split_sizes])
            rcv = (C.c_int64 * self.world)(*[int(s) * rb for s in output_split_sizes])
            return self._run(lambda st: self._ffi.check(self._lib.geogcn_comm_alltoallv(
                self._h, C.c_void_p(inp.data_ptr()), sb, C.c_void_p(out.data_ptr()), rcv, st), 'comm_alltoallv'),
                async_op, (out, inp))
        nbytes = inp.numel() * inp.element_size()
        assert out.is_contiguous() and inp.is_contiguous() and out.numel() * out.element_size() == nbytes and nbytes % self.world == 0
        return self._run(lambda st: self._ffi.check(self._lib.geogcn_comm_alltoall(
            self._h, C.c_void_p(inp.data_ptr()), C.c_void_p(out.data_ptr()), nbytes // self.world, st), 'comm_alltoall'),
            async_op, (out, inp))


class HostStagedGloo:
    """torch.distributed look-alike for ONE-GPU boxes (`GEOGCN_DIST_BACKEND=staged-gloo`): several ranks share cuda:0 --
    RCCL refuses that -- and every collective goes through a host copy and the gloo backend.  Slow and synchronous; it
    exists so that the REAL kernels can run under a REAL multi-rank partition (panels, narrow SpMM, slot layout, bf16 wire,
    ranks without rows) where no second GPU is available: a functional check, never a measurement."""

    class _Done:
        def wait(self):
            return True

    def __init__(self):
        import torch.distributed as dist
        self._d = dist
        self.ReduceOp = dist.ReduceOp

    @staticmethod
    def _bytes(t):
        return t.detach().contiguous().view(-1).view(torch.uint8).cpu()

    def _gather_bytes(self, t):
        ci = self._bytes(t)
        parts = [torch.empty_like(ci) for _ in range(self._d.get_world_size())]
        self._d.all_gather(parts, ci)
        return parts

    def all_reduce(self, t, op=None, group=None, async_op=False):
        c = t.detach().cpu()
        self._d.all_reduce(c, op=self._d.ReduceOp.SUM)
        t.copy_(c)
        return self._Done()

    def all_gather_into_tensor(self, out, inp, group=None, async_op=False):
        out.view(-1).view(torch.uint8).copy_(torch.cat(self._gather_bytes(inp)))
        return self._Done()

    def all_to_all_single(self, out, inp, output_split_sizes=None, input_split_sizes=None, group=None, async_op=False):
        if output_split_sizes is not None:
            # halo exchange: per-peer row counts; as bytes through gloo's own all-to-all-v
            rb = inp.shape[1] * inp.element_size()
            ci = self._bytes(inp)
            co = torch.empty(sum(int(s) for s in output_split_sizes) * rb, dtype=torch.uint8)
            self._d.all_to_all_single(co, ci, [int(s) * rb for s in output_split_sizes], [int(s) * rb for s in input_split_sizes])
            if co.numel():
                out.view(-1).view(torch.uint8).copy_(co)
            return self._Done()
        # (gloo's own all-to-all rejects some dtypes: every rank gathers every send buffer and keeps its own panel of each)
        w, r = self._d.get_world_size(), self._d.get_rank()
        parts = self._gather_bytes(inp)
        n = parts[0].numel() // w
        out.view(-1).view(torch.uint8).copy_(torch.cat([p[r * n:(r + 1) * n] for p in parts]))
        return self._Done()


def backend_name():
    """GEOGCN_DIST_BACKEND: 'torch' (default: torch.distributed's nccl = RCCL), 'native' (the library's geogcn_comm_* entry
    points for the data path), 'staged-gloo' (one-GPU functional check, see HostStagedGloo)."""
    return tuning.dist_backend()


def init_process_group(local_rank):
    """Join the job torchrun started (RANK / WORLD_SIZE / MASTER_* in the environment) -> this rank's device."""
    import torch.distributed as dist
    if backend_name() == 'staged-gloo':
        device = torch.device('cuda', 0)                     # all ranks share the one GPU
        torch.cuda.set_device(device)
        if not dist.is_initialized():
            dist.init_process_group('gloo')
        return device
    device = torch.device('cuda', int(local_rank))
    torch.cuda.set_device(device)
    if not dist.is_initialized():
        dist.init_process_group('nccl', device_id=device)
    return device


class TorchDistComm(Comm):
    """One process per GPU; RCCL (or gloo) through torch.distributed -- or, with GEOGCN_DIST_BACKEND=native, rendezvous
    through torch.distributed and the data path through the library's own RCCL entry points (NativeRccl); or
    GEOGCN_DIST_BACKEND=staged-gloo for the one-GPU functional check (HostStagedGloo)."""

    def __init__(self, N, device, group=None, exchange=None, balance=True):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        if backend_name() == 'staged-gloo' and torch.device(device).type == 'cuda':
            self.dist = HostStagedGloo()
        elif backend_name() == 'native' and torch.device(device).type == 'cuda':
            box = [NativeRccl.unique_id() if self.rank == 0 else None]
            dist.broadcast_object_list(box, src=0, group=group)
            self.dist = NativeRccl(self.world, self.rank, box[0], device)
        self.part = RowPartition(N, self.world, self.rank)
        self.device = device
        self.exchange = exchange or tuning.DIST_EXCHANGE
        self._auto = self.exchange == 'auto'      # prepare() may still move to 'halo' once it has seen the graph
        if self.exchange == 'auto':
            # at 2 ranks both schemes move the same bytes and the all-gather needs no repacking; from 3 ranks on
            # the all-to-all moves (w-1)/w * 2/w of what the all-gather delivers to every rank
            self.exchange = 'a2a' if self.world >= 3 else 'allgather'
        self._auto_default = self.exchange
        if self.exchange not in ('a2a', 'allgather', 'agpipe', 'halo'):
            raise ValueError("GEOGCN_DIST_EXCHANGE must be 'a2a', 'allgather', 'agpipe' or 'halo', got %r" % self.exchange)
        # 'agpipe': the all-gather scheme cut into FEATURE SLABS ("gather slab j+1 || SpMM slab j")
        self.slabs = max(1, int(tuning.DIST_AG_SLABS))
        self.balance = bool(balance)        # all-gather / halo schemes: cost-balanced row split (False: uniform, the A/B)
        self.halo = None                    # HaloPlan, built by graph_operand
        self.halo_rows = None               # rows received per exchange and rank under the halo scheme (prepare)
        self._bufs = {}

    @property
    def capturable(self):
        # Measured on the GPU box (tools/rccl_capture_probe.py, RCCL 2.26.6, world 1): all_reduce and all_gather_into_tensor -- blocking
        # or async -- are captured and replayed; all_to_all_single (a group of point-to-point transfers) is NOT safe under capture
        # (async: segmentation fault in capture_end; blocking: replays, then the communicator hangs at teardown).  So the all-gather
        # schemes only; a2a / halo steps stay eager.
        if self.exchange not in ('allgather', 'agpipe'):
            return False
        if isinstance(self.dist, HostStagedGloo) or torch.device(self.device).type != 'cuda':
            return False
        if isinstance(self.dist, NativeRccl):
            return True
        try:
            return self.dist.get_backend(self.group) == 'nccl'
        except Exception:
            return bool(getattr(self.dist, 'capturable', False))          # (a stand-in transport says so itself: tools/sim_rank.py)

    # -- row split ---------------------------------------------------------------------------------------
    def prepare(self, A_host):
        """all-gather scheme: re-cut the row blocks so that every rank gets the same cost (stored edges + per-row
        term) instead of the same number of rows.  Every rank computes the same cuts from the same host matrix.  The
        a2a scheme keeps the uniform split: there the SpMM work does not depend on the row split (every rank walks all of
        A_hat on its feature panel) and the dense work is proportional to rows."""
        if self.world <= 1:
            return
        A_csr = sps.csr_matrix(A_host)
        if A_csr.shape[0] != self.part.N:
            return
        bounds = balanced_bounds(A_csr.indptr, self.world) if self.balance else self.part.bounds_all
        if self._auto or self.exchange == 'halo':
            from . import graph
            self._symmetric = bool(graph.is_symmetric(A_csr))
            self.halo_rows = halo_sizes(A_csr, bounds, self._symmetric)
            remote = self.part.N - np.diff(bounds)
            limit = tuning.DIST_HALO_MAX_FRACTION_2_RANKS if self.world == 2 else tuning.DIST_HALO_MAX_FRACTION
            if self._auto:        # (decided per graph: a second graph without locality goes back to the default scheme)
                self.exchange = 'halo' if self.halo_rows.max() <= limit * max(1, remote.min()) else self._auto_default
                if self.exchange != 'halo':
                    self.halo = None
        if self.exchange in ('allgather', 'agpipe', 'halo') and self.balance:
            self.part = RowPartition(self.part.N, self.world, self.rank, bounds=bounds)
            self._bufs = {}
        elif not self.part.uniform:          # (a2a after an earlier graph had moved this communicator to a balanced split)
            self.part = RowPartition(self.part.N, self.world, self.rank)
            self._bufs = {}
        if self.exchange == 'halo':
            self._build_halo(A_csr)

    def _build_halo(self, A_csr):
        """The halo plan belongs to the GRAPH (prepare sees the whole adjacency); operands derived from it later -- the
        transpose restricted to the training columns -- reference a subset of the same rows and reuse the plan."""
        sym = getattr(self, '_symmetric', None)
        if sym is None:
            from . import graph
            sym = self._symmetric = bool(graph.is_symmetric(A_csr))
        self.halo = HaloPlan(self.part, *_pattern_rows(A_csr, self.part.r0, self.part.r1, sym))
        self._send_rows = torch.from_numpy(self.halo.send_rows).to(self.device)
        self._bufs = {}

    # -- constant operand ----------------------------------------------------------------------------
    def graph_operand(self, A_host):
        """A_hat as this exchange scheme needs it: the whole matrix (a2a) or the local row block (allgather), columns
        (and, for a2a, rows) in slot coordinates."""
        K = backend.active()
        part = self.part
        A_csr = sps.csr_matrix(A_host).astype(np.float32)
        A_csr.sort_indices()
        At = sps.csr_matrix(A_csr.T)
        At.sort_indices()
        if self.exchange == 'halo':
            if self.halo is None or self.halo.part is not part:
                self._build_halo(A_csr)             # (prepare was not called: this matrix IS the graph)
            Af, Ab = self.halo.local_rows_csr(A_csr), self.halo.local_rows_csr(At)
        elif self.exchange == 'a2a':
            Af, Ab = part.padded_square_csr(A_csr), part.padded_square_csr(At)
        else:
            Af = part.local_rows_csr(A_csr, part.n_gathered)
            Ab = part.local_rows_csr(At, part.n_gathered)
        same = (Af.shape == Ab.shape and np.array_equal(Af.indptr, Ab.indptr) and np.array_equal(Af.indices, Ab.indices)
                and np.array_equal(Af.data, Ab.data))
        # (slot positions are monotone in the global index: the stored order, hence the accumulation order, is unchanged;
        #  halo positions are not monotone -- `sort=False` keeps the stored order there)
        fwd = K.CSR(Af, self.device, sort=False)
        bwd = fwd if same else K.CSR(Ab, self.device, sort=False)
        return K.SparseOperand(fwd, bwd, same)

    # -- buffers ---------------------------------------------------------------------------------------
    def _gather_buffer(self, F, tag, bf16=False):
        K = backend.active()
        key = ('ag', int(F), tag, bool(bf16))
        buf = self._bufs.get(key)
        if buf is None:
            # zero once: the unused rows of a slot are never written and must read as 0
            if bf16:
                buf = K.HMat(self.part.n_gathered, F, self.device)
                buf.t.zero_()
            else:
                buf = K.DMat(self.part.n_gathered, F, self.device, ld=K.gather_ld(F))
            self._bufs[key] = buf
        return buf

    def _halo_buffers(self, F, tag, bf16=False):
        """(operand [own rows | halo], send buffer) of one exchange site; both keep the gather pitch."""
        K = backend.active()
        key = ('halo', int(F), tag, bool(bf16))
        bufs = self._bufs.get(key)
        if bufs is None:
            n = max(self.halo.n_operand, 1)
            op = K.HMat(n, F, self.device) if bf16 else K.DMat(n, F, self.device, ld=K.gather_ld(F))
            if bf16:
                op.t.zero_()
            send = torch.zeros((max(self.halo.n_send, 1), op.ld), dtype=op.t.dtype, device=self.device)
            bufs = self._bufs[key] = (op, send)
        return bufs

    def slab_width(self, F, bf16=False):
        """agpipe: columns per feature slab (whole float4s / bf16 octets)."""
        q = 8 if bf16 else 4
        return ((int(F) + self.slabs - 1) // self.slabs + q - 1) // q * q

    def _slab_buffers(self, F, tag, bf16=False):
        """agpipe: (send, recv) of one exchange site.  send = my rows as `slabs` feature panels [slab][R][ws] (the layout the
        GEMM's panel epilogue writes); recv[slab] = [world * R][ws], every rank's panel of that slab, one all-gather each."""
        K = backend.active()
        ws = self.slab_width(F, bf16)
        key = ('agp', ws, tag, bool(bf16))
        bufs = self._bufs.get(key)
        if bufs is None:
            send = K.Panels(self.part.n_local, self.slabs * ws, self.part.R, self.slabs, ws, self.device, bf16=bf16)
            recv = torch.zeros((self.slabs, self.part.n_gathered, ws), dtype=send.t.dtype, device=self.device)
            bufs = self._bufs[key] = (send, recv)
        bufs[0].F = int(F)
        return bufs

    def _panels(self, kind, F, tag, bf16=False):
        K = backend.active()
        wp = self.panel_width(F, bf16)
        key = (kind, wp, tag, bool(bf16))
        p = self._bufs.get(key)
        if p is None:
            p = self._bufs[key] = K.Panels(self.part.n_local, self.world * wp, self.part.R, self.world, wp, self.device, bf16=bf16)
        p.F = int(F)
        return p

    @staticmethod
    def _bf16(precision):
        return backend.active().bf16_gather(precision)

    def matmul_target(self, F, tag=None, precision=None, direct=True):
        """Where the product that feeds the exchange should write: my slot of the gathered buffer (allgather) or the
        all-to-all's send panels (a2a; `direct=False`: the producer is not a GEMM and needs a plain matrix, which
        graph_spmm_begin then packs)."""
        K = backend.active()
        bf16 = self._bf16(precision) and direct
        if self.exchange == 'halo':
            return self._halo_buffers(F, tag, bf16)[0].rows(0, self.part.n_local)    # the GEMM writes the head of the operand
        if self.exchange == 'allgather':
            buf = self._gather_buffer(F, tag, bf16)
            lo = self.rank * self.part.R
            return buf.rows(lo, lo + self.part.n_local)         # the GEMM writes straight into my slot
        if not direct:
            return K.DMat.empty(self.part.n_local, F, self.device)
        if self.exchange == 'agpipe':
            return self._slab_buffers(F, tag, bf16)[0]          # the GEMM writes the slab panels itself
        return self._panels('send', F, tag, bf16)

    def stage_operand(self, m, F, tag=None, precision=None):
        """The backward's dS (a plain fp32 matrix) as the exchange's operand."""
        K = backend.active()
        bf16 = self._bf16(precision)
        if self.exchange in ('allgather', 'halo'):
            g = self.matmul_target(F, tag=tag, precision=precision)
            if bf16:
                K.cast_bf16(m, out=g)
            else:
                g.copy_from(m)
            return g
        send = self._slab_buffers(F, tag, bf16)[0] if self.exchange == 'agpipe' else self._panels('send', F, tag, bf16)
        if bf16:
            # fp32 staging buffer with the bf16 panels' geometry (their width is a multiple of 8, not of 4)
            key = ('stage', send.wp, send.W, tag)
            stage = self._bufs.get(key)
            if stage is None:
                stage = self._bufs[key] = torch.zeros(send.t.numel(), dtype=torch.float32, device=self.device)
            K.pack_panels(m, self.part.R, send.W, send.wp, stage)
            K.cast_bf16_flat(stage, send.t, send.wp)
        else:
            K.pack_panels(m, self.part.R, send.W, send.wp, send.t)
        return send

    # -- the exchange around A . Z ----------------------------------------------------------------------
    def panel_width(self, F, bf16=False):
        q = 8 if bf16 else 4
        return ((int(F) + self.world - 1) // self.world + q - 1) // q * q

    def graph_spmm_begin(self, A_csr, z, bias, act, F, tag=None):
        """Start act(A . Z + bias) for row-partitioned Z: the exchange of Z is issued asynchronously (RCCL runs it
        on its own stream); continue with graph_spmm_mid / graph_spmm_end."""
        K = backend.active()
        h = dict(A=A_csr, bias=bias, act=act, F=F, tag=tag, mid=False)
        if self.exchange == 'halo':
            op, send = self._halo_buffers(F, tag, isinstance(z, K.HMat))
            hp = self.halo
            K.pack_rows(op.rows(0, self.part.n_local), self._send_rows, send[:hp.n_send])
            h['buf'] = op
            h['work'] = self.dist.all_to_all_single(op.t[self.part.n_local:self.part.n_local + hp.n_halo], send[:hp.n_send],
                                                    output_split_sizes=hp.recv_counts.tolist(),
                                                    input_split_sizes=hp.send_counts.tolist(), group=self.group, async_op=True)
            return h
        if self.exchange == 'allgather':
            buf = self._gather_buffer(F, tag, isinstance(z, K.HMat))
            R = self.part.R
            h['buf'] = buf
            h['work'] = self.dist.all_gather_into_tensor(buf.t, buf.t[self.rank * R:(self.rank + 1) * R], group=self.group,
                                                         async_op=True)
            return h
        if self.exchange == 'agpipe':
            send, recv = self._slab_buffers(F, tag, isinstance(z, K.Panels) and z.bf16)
            if not isinstance(z, K.Panels):                      # a producer that could not write panels itself
                K.pack_panels(z, self.part.R, send.W, send.wp, send.t)
            elif z is not send:
                raise ValueError("agpipe: the operand must be the panels matmul_target / stage_operand handed out")
            R, ws = self.part.R, send.wp
            # one all-gather per slab, issued back to back (RCCL runs them in order on its own stream): slab q + 1 is on the
            # wire while graph_spmm_end multiplies slab q
            h.update(recv=recv, ws=ws, bf16=send.bf16,
                     works=[self.dist.all_gather_into_tensor(recv[q].view(-1), send.t[q * R * ws:(q + 1) * R * ws], group=self.group,
                                                             async_op=True) for q in range(send.W)])
            h['work'] = None
            return h
        if not isinstance(z, K.Panels):                          # a producer that could not write panels itself
            send = self._panels('send', F, tag, False)
            K.pack_panels(z, self.part.R, self.world, send.wp, send.t)
            z = send
        recv = self._panels('recv', F, tag, z.bf16)
        h.update(send=z, recv=recv, wp=z.wp)
        h['work'] = self.dist.all_to_all_single(recv.t, z.t, group=self.group, async_op=True)
        return h

    def graph_spmm_mid(self, h):
        """a2a: wait for panel `rank` of all rows, multiply, start the return exchange.  allgather: nothing to do."""
        if h['mid'] or self.exchange in ('allgather', 'agpipe', 'halo'):
            return h
        K = backend.active()
        part = self.part
        h['mid'] = True
        if h['work'] is not None:
            h['work'].wait()            # nccl: the compute stream waits for the collective; gloo: the host does
        F, wp = h['F'], h['wp']
        sout = self._bufs.get(('sout', wp, h['tag']))
        if sout is None:
            sout = self._bufs[('sout', wp, h['tag'])] = K.DMat(part.n_gathered, wp, self.device, ld=wp)
        # my panel holds columns [rank*wp, rank*wp + Fq) of the layer (the last panels may be partly / wholly padding)
        Fq = max(0, min(wp, F - self.rank * wp))
        if Fq > 0:
            bias = h['bias']
            bslab = None if bias is None else bias[self.rank * wp: self.rank * wp + K.pad4(Fq)]
            K.spmm(h['A'], h['recv'].as_rows(), out=sout, bias=bslab, act=h['act'], F=Fq)
        back = self._bufs.get(('back', wp, h['tag']))
        if back is None:
            back = self._bufs[('back', wp, h['tag'])] = torch.zeros(sout.t.numel(), dtype=torch.float32, device=self.device)
        h['back'] = back
        h['work2'] = self.dist.all_to_all_single(back, sout.t.view(-1), group=self.group, async_op=True)
        return h

    def graph_spmm_end(self, h):
        """-> row-partitioned result (n_local x F)."""
        K = backend.active()
        if self.exchange in ('allgather', 'halo'):
            if h['work'] is not None:
                h['work'].wait()
            return K.spmm(h['A'], h['buf'], bias=h['bias'], act=h['act'], F=h['F'])
        if self.exchange == 'agpipe':
            F, ws = h['F'], h['ws']
            out = K.DMat.empty(self.part.n_local, F, self.device)
            for q, work in enumerate(h['works']):
                if work is not None:
                    work.wait()                                  # the compute stream waits for slab q only
                Fq = max(0, min(ws, F - q * ws))
                if Fq == 0:
                    continue
                v = h['recv'][q]
                B = K.HMat(v.shape[0], ws, t=v) if h['bf16'] else K.DMat(v.shape[0], ws, t=v)
                bias = h['bias']
                K.spmm(h['A'], B, out=out, out_col0=q * ws, bias=None if bias is None else bias[q * ws: q * ws + K.pad4(Fq)],
                       act=h['act'], F=Fq)
            return out
        self.graph_spmm_mid(h)
        h['work2'].wait()
        out = K.DMat.empty(self.part.n_local, h['F'], self.device)
        K.unpack_panels(h['back'], self.part.R, self.world, h['wp'], out)
        return out

    def all_gather_rows(self, local):
        """(n_local x F) -> (N x F) on every rank, rows in GLOBAL order."""
        K = backend.active()
        buf = self._gather_buffer(local.F, 'out')
        R = self.part.R
        lo = self.rank * R
        buf.rows(lo, lo + self.part.n_local).copy_from(local)
        self.dist.all_gather_into_tensor(buf.t, buf.t[lo:lo + R], group=self.group)
        if self.part.uniform:
            return buf
        pos = torch.from_numpy(self.part.slot_position(np.arange(self.part.N))).to(buf.t.device)
        return K.DMat(self.part.N, local.F, t=buf.t.index_select(0, pos).contiguous())

    def all_reduce_sum_(self, t: torch.Tensor):
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return t
