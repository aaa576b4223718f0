"""The step right before the hot path: the normalised adjacency from an edge list, its transpose
check, and an optional node reordering.

Reference: gcnmain.py:115-128 builds A_hat from a networkx graph --
    adj = adjacency_matrix(graph, weight='w')      (no 'w' attribute is ever set: unit weights, data.py:56,61)
    adj.setdiag(0); adj.setdiag(1)                 (self loops of weight 1)
    A_hat = D^-1/2 . adj . D^-1/2                  (float64, 1/sqrt(0) -> 0), then .astype(float32)
``build_ahat`` does the same from an (m x 2) edge array.

Reordering.  The graph product S = A_hat . Z gathers one 1.2 KB row of Z per stored edge; whether those gathers hit the
4 MB L2 of the XCD that issues them depends only on WHERE in memory the neighbours of nearby rows live, i.e. on the node
numbering.  A numbering that puts tightly connected nodes next to each other (reverse Cuthill-McKee, breadth-first
order) turns the random gather of a community-structured graph into a sliding window; on a graph without such
structure (the pinned power-law generator: neighbours are drawn independently of the node index) no numbering can.
``Reordering`` carries the permutation both ways so that callers keep talking in ORIGINAL node ids: GraphConv
(``reorder=...``) permutes A_hat, X, the index vectors and the injected dropout mask on the way in and every per-node
output (probabilities, gates) on the way out."""
from __future__ import annotations

import numpy as np
import scipy.sparse as sps
from scipy.sparse import csgraph

from .synth import normalize_adjacency

REORDERINGS = (None, 'none', 'degree', 'rcm', 'bfs', 'lpa', 'auto')


def adjacency_from_edges(edges, N: int) -> sps.csr_matrix:
    """Symmetric 0/1 adjacency with unit self loops from an (m x 2) array of node pairs (duplicates and both
    orientations collapse; the reference's graph is undirected and unweighted)."""
    edges = np.asarray(edges, dtype=np.int64).reshape(-1, 2)
    if edges.size and (edges.min() < 0 or edges.max() >= N):
        raise IndexError("edge endpoint outside [0, %d)" % N)
    r, c = edges[:, 0], edges[:, 1]
    keep = r != c                                            # setdiag(0) first: self loops in the input are dropped ...
    r, c = r[keep], c[keep]
    A = sps.coo_matrix((np.ones(2 * len(r), dtype=np.int64), (np.r_[r, c], np.r_[c, r])), shape=(N, N)).tocsr()
    A.data[:] = 1
    return (A + sps.identity(N, dtype=np.int64, format='csr')).tocsr()       # ... then setdiag(1)


class Reordering:
    """perm[new] = old node id; inv[old] = new position."""

    def __init__(self, perm):
        self.perm = np.ascontiguousarray(perm, dtype=np.int64)
        self.inv = np.empty_like(self.perm)
        self.inv[self.perm] = np.arange(len(self.perm), dtype=np.int64)

    def matrix(self, A: sps.spmatrix) -> sps.csr_matrix:
        """P A P^T: rows and columns renumbered (values untouched: the same fp32 numbers in new places)."""
        A = sps.csr_matrix(A)
        B = sps.csr_matrix(A[self.perm][:, self.perm])
        B.sort_indices()
        B.indptr = B.indptr.astype(np.int32)
        B.indices = B.indices.astype(np.int32)
        return B

    def rows(self, M):
        """Per-node rows (X, Y, a dropout mask) into the new order."""
        return M[self.perm]

    def indices(self, idx):
        """Original node ids -> positions in the new order."""
        return self.inv[np.asarray(idx, dtype=np.int64)]

    def restore_rows(self, M):
        """Per-node rows computed in the new order back to original node order."""
        return M[self.inv]


def label_propagation(A: sps.spmatrix, iters: int = 12) -> np.ndarray:
    """Community labels by (semi-synchronous) label propagation: every node repeatedly adopts the label most frequent
    among its neighbours (ties: the smallest label); odd and even nodes move in alternate sweeps so that the synchronous
    update cannot oscillate.  Vectorised: one sort of the (row, neighbour label) pairs per sweep, O(E log E).  Unlike
    breadth-first / Cuthill-McKee orders -- which on a small-world graph reach everything within three levels and end up
    ordering by distance from the hub -- this finds the dense groups themselves."""
    A = sps.csr_matrix(A)
    N = A.shape[0]
    indices = A.indices.astype(np.int64)
    row_of = np.repeat(np.arange(N, dtype=np.int64), np.diff(A.indptr))
    labels = np.arange(N, dtype=np.int64)
    for it in range(iters):
        key = row_of * N + labels[indices]
        key.sort(kind='stable')
        starts = np.flatnonzero(np.r_[True, key[1:] != key[:-1]])
        counts = np.diff(np.r_[starts, len(key)])
        rows, labs = key[starts] // N, key[starts] % N
        o = np.lexsort((labs, -counts, rows))
        first = np.flatnonzero(np.r_[True, rows[o][1:] != rows[o][:-1]])
        best_rows, best = rows[o][first], labs[o][first]
        move = (best_rows + it) % 2 == 0
        new = labels.copy()
        new[best_rows[move]] = best[move]
        if np.array_equal(new, labels) and it >= 2:
            break
        labels = new
    return labels


def reordering(A: sps.spmatrix, method) -> Reordering | None:
    """Node permutation of the (symmetric-pattern) adjacency `A`:
      'degree'  decreasing stored-edge count (hubs first; a locality-free baseline that only groups the hot rows),
      'rcm'     reverse Cuthill-McKee (bandwidth reduction: neighbours end up close in index),
      'bfs'     breadth-first order, components in order of their highest-degree node,
      'lpa'     communities found by label propagation, largest first, hubs first inside each,
      'auto'    'lpa' if that at least doubles the share of edges inside an L2 window of their row (to >= 30 %), else none."""
    if method in (None, 'none'):
        return None
    if method == 'auto':
        # label propagation, kept only when it pays: the share of edges inside one L2 window of their row must at least
        # double and reach 30 % (a graph without communities -- the pinned power-law generator -- stays as it is)
        before = locality_profile(A)['within_window']
        ro = reordering(A, 'lpa')
        after = locality_profile(ro.matrix(A))['within_window']
        return ro if after >= max(0.3, 2.0 * before) else None
    A = sps.csr_matrix(A)
    N = A.shape[0]
    deg = np.diff(A.indptr)
    if method == 'degree':
        perm = np.argsort(-deg, kind='stable')
    elif method == 'rcm':
        perm = csgraph.reverse_cuthill_mckee(sps.csr_matrix((np.ones(A.nnz, np.int8), A.indices, A.indptr), shape=A.shape),
                                             symmetric_mode=True)
    elif method == 'bfs':
        pattern = sps.csr_matrix((np.ones(A.nnz, np.int8), A.indices, A.indptr), shape=A.shape)
        seen = np.zeros(N, dtype=bool)
        out = []
        for start in np.argsort(-deg, kind='stable'):
            if seen[start]:
                continue
            order = csgraph.breadth_first_order(pattern, int(start), directed=False, return_predecessors=False)
            order = order[~seen[order]]
            seen[order] = True
            out.append(order)
            if seen.all():
                break
        perm = np.concatenate(out)
    elif method == 'lpa':
        labels = label_propagation(A)
        # communities in order of size (largest first), hubs first inside a community
        size = np.bincount(labels, minlength=N)
        perm = np.lexsort((-deg, labels, -size[labels]))
    else:
        raise ValueError("reorder must be one of %r, got %r" % (REORDERINGS, method))
    return Reordering(perm)


def build_ahat(edges, N: int, reorder=None, dtype=np.float32):
    """(A_hat, reordering) from an edge list: A_hat = D^-1/2 (A + I) D^-1/2 as the reference computes it
    (gcnmain.py:115-128), float64 then cast, int32 CSR with sorted indices; with `reorder`, renumbered (the returned
    Reordering maps between the two numberings; None when reorder is None)."""
    Ah = normalize_adjacency(adjacency_from_edges(edges, N), dtype=dtype)
    ro = reordering(Ah, reorder)
    return (Ah if ro is None else ro.matrix(Ah)), ro


def is_symmetric(A: sps.spmatrix) -> bool:
    """Exact symmetry of values and pattern (true for the normalised adjacency of an unweighted graph, so
    A^T . G may reuse A's CSR)."""
    A = sps.csr_matrix(A)
    if A.shape[0] != A.shape[1]:
        return False
    At = sps.csr_matrix(A.T)
    At.sort_indices()
    B = A.copy()
    B.sort_indices()
    return bool(np.array_equal(At.indptr, B.indptr) and np.array_equal(At.indices, B.indices) and np.array_equal(At.data, B.data))


def locality_profile(A: sps.spmatrix, window_rows: int = 3276):
    """How local the gathers of A . Z are under the current numbering: the fraction of stored edges whose column lies
    within `window_rows` of the row (3276 rows x 1280 B = the 4 MB of one XCD's L2), and the mean |row - col|."""
    A = sps.csr_matrix(A)
    row_of = np.repeat(np.arange(A.shape[0], dtype=np.int64), np.diff(A.indptr))
    d = np.abs(row_of - A.indices)
    return {'within_window': float((d <= window_rows).mean()) if len(d) else 1.0, 'mean_distance': float(d.mean()) if len(d) else 0.0}
