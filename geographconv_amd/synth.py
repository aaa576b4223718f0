"""Pinned synthetic inputs for the GCN hot path.

The reference ships no data (raw tweets withdrawn, README.md:24) and no generator, so every
parity test, the CPU baseline and bench.py draw (A_hat, X, Y) from the two generators below.
They mirror how the reference builds its inputs:

* ``powerlaw_ahat``  -- A_hat = D^-1/2 (A + I) D^-1/2 computed in float64 then cast to float32
  CSR with int32 indices (reference gcnmain.py:115-128; unit edge weights because no 'w'
  attribute is ever set, data.py:56,61).
* ``bow_x``          -- binary-tf x smooth-idf, row-L2-normalised, float32 CSR bag of words
  (reference data.py:264-285: TfidfVectorizer(binary=True, norm='l2', dtype=float32)).

Both use numpy's legacy ``RandomState`` so the streams are stable; ``PINNED`` holds the
CRC32s of the index arrays measured when SURVEY.md was written.  ``check_pinned`` detects a
numpy/scipy that changed the streams (regenerate + re-pin in that case, do not "fix" tests).
"""
from __future__ import annotations

import zlib
from dataclasses import dataclass

import numpy as np
import scipy.sparse as sps


def powerlaw_ahat(N: int, E_target: int, alpha: float = 0.8, seed: int = 0) -> sps.csr_matrix:
    """Symmetric power-law graph -> normalised adjacency (float32 CSR, int32, sorted)."""
    rng = np.random.RandomState(seed)
    w = np.arange(1, N + 1, dtype=np.float64) ** -alpha
    rng.shuffle(w)
    p = w / w.sum()
    m = int((E_target - N) // 2 * 1.08)
    r = rng.choice(N, size=m, p=p)
    c = rng.randint(0, N, size=m)
    keep = r != c
    r, c = r[keep], c[keep]
    A = sps.coo_matrix((np.ones(2 * len(r), dtype=np.int64), (np.r_[r, c], np.r_[c, r])),
                       shape=(N, N)).tocsr()
    A.data[:] = 1                                   # binarise (tocsr summed duplicates)
    A = (A + sps.identity(N, dtype=np.int64, format='csr')).tocsr()   # self loop = 1
    return normalize_adjacency(A)


def community_edges(N: int, E_target: int, n_comm: int, p_in: float = 0.9, alpha: float = 0.8, seed: int = 0):
    """Edge list of a graph with the pinned generator's size and degree skew AND community structure: nodes belong to
    `n_comm` equal communities (membership shuffled over the node ids, so the given numbering has no locality at all);
    each of the m draws picks its first endpoint with the power-law weights of `powerlaw_ahat` and its second endpoint
    inside the first one's community with probability p_in, anywhere otherwise.  -> (edges (m x 2) int64, community id
    per node).  Used to show what the graph product does when the gather CAN be made local (geographconv_amd.graph)."""
    rng = np.random.RandomState(seed)
    w = np.arange(1, N + 1, dtype=np.float64) ** -alpha
    rng.shuffle(w)
    p = w / w.sum()
    comm = rng.permutation(N) % n_comm                       # community of node i
    order = np.argsort(comm, kind='stable')                  # members of community k: order[start[k]:start[k+1]]
    start = np.concatenate([[0], np.cumsum(np.bincount(comm, minlength=n_comm))])
    # duplicates inside a community collapse when the adjacency is binarised: oversample so that the stored count lands
    # near E_target (factor fitted once for the pinned call below)
    m = int((E_target - N) // 2 * 1.30)
    r = rng.choice(N, size=m, p=p)
    inside = rng.rand(m) < p_in
    k = comm[r]
    c_in = order[start[k] + (rng.rand(m) * (start[k + 1] - start[k])).astype(np.int64)]
    c = np.where(inside, c_in, rng.randint(0, N, size=m))
    keep = r != c
    return np.stack([r[keep], c[keep]], axis=1), comm


def community_ahat(N: int, E_target: int, n_comm: int, p_in: float = 0.9, alpha: float = 0.8, seed: int = 0) -> sps.csr_matrix:
    """Normalised adjacency (gcnmain.py:115-128) of `community_edges`."""
    edges, _ = community_edges(N, E_target, n_comm, p_in, alpha, seed)
    r, c = edges[:, 0], edges[:, 1]
    A = sps.coo_matrix((np.ones(2 * len(r), dtype=np.int64), (np.r_[r, c], np.r_[c, r])), shape=(N, N)).tocsr()
    A.data[:] = 1
    A = (A + sps.identity(N, dtype=np.int64, format='csr')).tocsr()
    return normalize_adjacency(A)


def normalize_adjacency(A: sps.spmatrix, dtype=np.float32) -> sps.csr_matrix:
    """D^-1/2 A D^-1/2 in float64, then cast (reference gcnmain.py:121-128).

    ``A`` must already carry its self loops (gcnmain.py:117-120 sets the diagonal to 1).
    Zero-degree rows give 1/sqrt(0)=inf which the reference replaces by 0 (gcnmain.py:125).
    """
    A = sps.csr_matrix(A)
    d = np.asarray(A.sum(axis=1)).ravel().astype(np.float64)
    with np.errstate(divide='ignore'):
        ds = 1.0 / np.sqrt(d)
    ds[np.isinf(ds)] = 0.0
    D = sps.diags(ds).tocsr()
    Ah = (D @ A.astype(np.float64) @ D).astype(dtype).tocsr()
    Ah.sort_indices()
    Ah.indptr = Ah.indptr.astype(np.int32)
    Ah.indices = Ah.indices.astype(np.int32)
    return Ah


def bow_x(N: int, V: int, mean_nnz: float, seed: int = 1) -> sps.csr_matrix:
    """Zipfian bag-of-words, binary tf x smooth idf, L2 rows, float32 CSR."""
    rng = np.random.RandomState(seed)
    k = np.clip(rng.lognormal(np.log(mean_nnz) - 0.5 * 0.6 ** 2, 0.6, size=N).astype(np.int64),
                1, V // 4)
    pw = np.arange(1, V + 1, dtype=np.float64) ** -1.0
    pw /= pw.sum()
    rows = np.repeat(np.arange(N), k)
    cols = rng.choice(V, size=int(k.sum()), p=pw)
    X = sps.coo_matrix((np.ones(len(rows), dtype=np.float64), (rows, cols)), shape=(N, V)).tocsr()
    X.data[:] = 1.0
    df = np.bincount(X.indices, minlength=V).astype(np.float64)
    idf = np.log((1.0 + N) / (1.0 + df)) + 1.0
    X = X @ sps.diags(idf)
    rn = np.sqrt(np.asarray(X.multiply(X).sum(axis=1)).ravel())
    rn[rn == 0] = 1
    X = (sps.diags(1.0 / rn) @ X).astype(np.float32).tocsr()
    X.sort_indices()
    X.indptr = X.indptr.astype(np.int32)
    X.indices = X.indices.astype(np.int32)
    return X


def labels(N: int, C: int, seed: int = 2) -> np.ndarray:
    return np.random.RandomState(seed).randint(0, C, N).astype(np.int32)


def split_indices(N: int):
    """train = first 60 %, dev next 20 %, test last 20 % (reference layout gcnmain.py:189,211-212)."""
    n_tr = int(N * 0.6)
    n_dev = int(N * 0.2)
    tr = np.arange(0, n_tr, dtype=np.int32)
    dev = np.arange(n_tr, n_tr + n_dev, dtype=np.int32)
    te = np.arange(n_tr + n_dev, N, dtype=np.int32)
    return tr, dev, te


@dataclass(frozen=True)
class Shape:
    name: str
    N: int
    E_target: int
    V: int
    mean_nnz: int
    C: int


CMU = Shape('cmu', 9475, 130_000, 9_500, 100, 129)
TWUS = Shape('twus', 440_000, 10_000_000, 10_000, 64, 256)
SHAPES = {'cmu': CMU, 'twus': TWUS}

# (nnz, crc32(indptr), crc32(indices)) measured in the survey container
PINNED = {
    ('cmu', 'A'): (138_117, 0xe923e39d, 0x6567e474),
    ('cmu', 'X'): (685_615, 0x6c70f752, 0x6611042e),
    ('twus', 'A'): (10_730_596, 0x38c7a66c, 0x8601fd3c),
    ('twus', 'X'): (21_458_408, 0xdf8585c4, 0x3542a305),
    # community_ahat(440000, 10_000_000, 220): degree mean 25.84 / median 20 / p99 131 / max 11,833 / min 3 (numpy 2.2.6)
    ('twus_sbm', 'A'): (11_368_850, 0x219cf29c, 0x27282978),
}


def fingerprint(M: sps.csr_matrix):
    return (int(M.nnz), zlib.crc32(M.indptr.tobytes()), zlib.crc32(M.indices.tobytes()))


def check_pinned(shape: str, which: str, M: sps.csr_matrix) -> bool:
    return fingerprint(M) == PINNED[(shape, which)]


# TwitterUS size with community structure: 220 communities of 2,000 nodes, 90 % of the draws inside the community
TWUS_SBM_COMMUNITIES = 220


def make_graph(shape: str):
    """(A_hat, X, Y, (train_idx, dev_idx, test_idx), C) for 'cmu', 'twus' or 'twus_sbm' (TwitterUS size and skew with
    community structure; same X, Y and split as 'twus')."""
    s = SHAPES['twus' if shape == 'twus_sbm' else shape]
    if shape == 'twus_sbm':
        A = community_ahat(s.N, s.E_target, TWUS_SBM_COMMUNITIES)
    else:
        A = powerlaw_ahat(s.N, s.E_target)
    X = bow_x(s.N, s.V, s.mean_nnz)
    Y = labels(s.N, s.C)
    return A, X, Y, split_indices(s.N), s.C


def small_graph(N: int, avg_deg: float, V: int, mean_nnz: float, C: int, seed: int = 0,
                hub: bool = True, empty_rows: int = 0):
    """Small test graph with the awkward cases the tests want: a hub row and, optionally,
    isolated nodes (whose A_hat row holds only the self loop) / X rows that are empty."""
    A = powerlaw_ahat(N, max(int(N * avg_deg), N + 4), seed=seed)
    X = bow_x(N, V, mean_nnz, seed=seed + 1).tolil()
    for i in range(min(empty_rows, N)):
        X.rows[N - 1 - i] = []
        X.data[N - 1 - i] = []
    X = sps.csr_matrix(X, dtype=np.float32)
    X.sort_indices()
    X.indptr = X.indptr.astype(np.int32)
    X.indices = X.indices.astype(np.int32)
    Y = labels(N, C, seed=seed + 2)
    return A, X, Y
