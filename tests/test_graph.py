"""The step before the hot path: A_hat from an edge list as the reference computes it
(gcnmain.py:115-128), exact-symmetry check, node reorderings and their round trips."""
import numpy as np
import pytest
import scipy.sparse as sps

from geographconv_amd import graph, synth


def test_build_ahat_matches_the_reference_formula():
    rng = np.random.RandomState(0)
    N = 300
    edges = rng.randint(0, N, size=(1500, 2))
    edges = np.vstack([edges, edges[:100, ::-1], [[5, 5], [7, 7]]])        # duplicates, both orientations, self loops
    Ah, ro = graph.build_ahat(edges, N)
    assert ro is None and Ah.dtype == np.float32 and Ah.indices.dtype == np.int32 and Ah.has_sorted_indices
    # the formula, written out densely in float64
    D = np.zeros((N, N))
    for r, c in edges:
        if r != c:
            D[r, c] = D[c, r] = 1.0
    np.fill_diagonal(D, 1.0)
    d = D.sum(1)
    ref = (D / np.sqrt(d)[:, None] / np.sqrt(d)[None, :]).astype(np.float32)
    assert np.array_equal(Ah.toarray(), ref)
    assert graph.is_symmetric(Ah)                                       # exact in fp32 for unit weights
    asym = sps.csr_matrix(sps.diags(1.0 / d) @ sps.csr_matrix(D), dtype=np.float32)
    assert not graph.is_symmetric(asym)
    with pytest.raises(IndexError):
        graph.build_ahat([[0, N]], N)
    # an isolated node: degree 1 (its self loop) -> A_hat[i, i] = 1; the generator's formula and this one agree
    Ah2, _ = graph.build_ahat(np.zeros((0, 2), int), 4)
    assert np.array_equal(Ah2.toarray(), np.eye(4, dtype=np.float32))


@pytest.mark.parametrize("method", ['degree', 'rcm', 'bfs', 'lpa'])
def test_reorderings_are_permutations_and_commute_with_the_product(method):
    A = synth.powerlaw_ahat(800, 9000, seed=2)
    ro = graph.reordering(A, method)
    assert sorted(ro.perm.tolist()) == list(range(800))
    assert np.array_equal(ro.inv[ro.perm], np.arange(800))
    B = ro.matrix(A)
    assert B.nnz == A.nnz and graph.is_symmetric(B) and B.has_sorted_indices
    Z = np.random.RandomState(1).randn(800, 5)
    S = A.astype(np.float64) @ Z
    S2 = ro.restore_rows(B.astype(np.float64) @ ro.rows(Z))
    assert np.allclose(S, S2, rtol=1e-12, atol=1e-12)
    idx = np.array([3, 799, 0, 41])
    assert np.array_equal(ro.rows(np.arange(800))[ro.indices(idx)], idx)
    assert graph.reordering(A, None) is None and graph.reordering(A, 'none') is None
    with pytest.raises(ValueError):
        graph.reordering(A, 'alphabetical')


def test_label_propagation_recovers_planted_communities_and_reordering_makes_them_local():
    edges, comm = synth.community_edges(6000, 90000, n_comm=12, p_in=0.9, seed=3)
    Ah, ro = graph.build_ahat(edges, 6000, reorder='lpa')
    base, _ = graph.build_ahat(edges, 6000)
    lab = graph.label_propagation(base)
    # every found label is (almost) pure in the planted communities
    purity = sum(np.bincount(comm[lab == l]).max() for l in np.unique(lab)) / 6000.0
    assert purity > 0.95
    before = graph.locality_profile(base, window_rows=600)['within_window']
    after = graph.locality_profile(Ah, window_rows=600)['within_window']
    assert before < 0.25 and after > 0.8, (before, after)
    # bfs / rcm on a small-world graph do not (documented negative result)
    assert graph.locality_profile(graph.reordering(base, 'rcm').matrix(base), window_rows=600)['within_window'] < after


def test_auto_reordering_adopts_label_propagation_only_when_it_pays():
    """'auto' = label propagation if it at least doubles the share of edges inside an L2 window (to >= 30 %), else nothing:
    a community graph with shuffled ids gets reordered, the structure-free power-law generator does not."""
    from geographconv_amd import graph, synth
    A = synth.community_ahat(40000, 600000, 20, seed=3)
    ro = graph.reordering(A, 'auto')
    assert ro is not None
    assert graph.locality_profile(ro.matrix(A))['within_window'] >= 2 * graph.locality_profile(A)['within_window']
    B = synth.powerlaw_ahat(40000, 600000, seed=3)
    assert graph.reordering(B, 'auto') is None
