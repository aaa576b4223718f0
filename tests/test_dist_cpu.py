"""N>1 path on CPU: world_size-2 gloo run of GraphConv (row partition + in-place all-gather +
gradient all-reduce) with the NumPy test double standing in for the HIP kernels; results must
equal the single-process oracle on the same inputs.  Plus single-process checks of the partition
arithmetic."""
import os
import socket

import numpy as np
import pytest
import scipy.sparse as sps

from geographconv_amd import synth
from geographconv_amd.dist import RowPartition
from oracle import gcn_oracle as O


def test_row_partition_covers_rows_once():
    for N, w in [(10, 3), (440000, 8), (7, 8), (16, 4)]:
        seen = np.zeros(N, int)
        for r in range(w):
            p = RowPartition(N, w, r)
            seen[p.r0:p.r1] += 1
            assert p.n_gathered >= N and p.n_gathered == p.R * w
            assert p.r0 == min(N, r * p.R)
        assert np.all(seen == 1)


def test_padded_square_csr():
    A = sps.random(10, 10, density=0.3, format='csr', dtype=np.float32, random_state=1)
    p = RowPartition(10, 3, 0)                    # R = 4, gathered = 12
    B = p.padded_square_csr(A)
    assert B.shape == (12, 12) and B.nnz == A.nnz and (B[:10, :10] != A).nnz == 0
    assert B[10:].nnz == 0


def test_partition_split_indices_and_csr_padding():
    p = RowPartition(10, 2, 1)             # rows 5..9
    idx = np.array([9, 1, 5, 4, 7], dtype=np.int32)
    y = np.array([0, 1, 2, 3, 4], dtype=np.int32)
    loc, yl, sel = p.split_indices(idx, y)
    assert list(loc) == [4, 0, 2] and list(yl) == [0, 2, 4] and sel.sum() == 3
    A = sps.random(10, 10, density=0.3, format='csr', dtype=np.float32, random_state=0)
    blk = RowPartition(10, 3, 2).local_rows_csr(A, 12)
    assert blk.shape == (2, 12) and (blk[:, :10] != A[8:10]).nnz == 0


def test_cost_balanced_bounds_and_slot_layout():
    """all-gather scheme: row blocks cut by cost (stored edges + a per-row term); the gathered buffer gives every rank a
    slot of R = max block rows, and A's columns are renumbered to slot positions WITHOUT changing the stored order."""
    from geographconv_amd.dist import balanced_bounds
    A = synth.powerlaw_ahat(3000, 40000, seed=3)
    w = 4
    b = balanced_bounds(A.indptr, w, row_cost=10.0)
    assert b[0] == 0 and b[-1] == 3000 and np.all(np.diff(b) > 0)
    cost = np.diff(A.indptr) + 10.0
    per = [cost[b[r]:b[r + 1]].sum() for r in range(w)]
    assert max(per) <= 1.05 * (cost.sum() / w) + cost.max()              # balanced up to one (hub) row
    rows = [b[r + 1] - b[r] for r in range(w)]
    assert max(rows) > min(rows)                                        # not the uniform split
    parts = [RowPartition(3000, w, r, bounds=b) for r in range(w)]
    R = parts[0].R
    assert R == max(rows) and parts[0].n_gathered == R * w
    seen = np.zeros(3000, int)
    for p in parts:
        seen[p.r0:p.r1] += 1
        pos = p.slot_position(np.arange(p.r0, p.r1))
        assert np.array_equal(pos, p.rank * R + np.arange(p.n_local))
        blk = p.local_rows_csr(A, p.n_gathered)
        ref = A[p.r0:p.r1]
        assert blk.shape == (p.n_local, R * w) and np.array_equal(blk.indptr, ref.indptr) and np.array_equal(blk.data, ref.data)
        assert np.array_equal(blk.indices, p.slot_position(ref.indices))
        assert blk.has_sorted_indices                                   # slot positions are monotone in the global index
    assert np.all(seen == 1)
    # the product through the slot layout equals the plain one
    Z = np.random.RandomState(0).randn(3000, 7).astype(np.float32)
    G = np.zeros((R * w, 7), np.float32)
    G[parts[0].slot_position(np.arange(3000))] = Z
    for p in parts:
        assert np.array_equal(p.local_rows_csr(A, p.n_gathered) @ G, A[p.r0:p.r1] @ Z)
    sq = parts[1].padded_square_csr(A)
    S = sq @ G
    assert np.array_equal(S[parts[0].slot_position(np.arange(3000))], A @ Z)


@pytest.mark.parametrize("symmetric", [True, False])
def test_halo_plan_lists_agree_across_ranks_and_keep_the_product(symmetric):
    """halo scheme: every rank derives its send and receive lists from ITS OWN rows; they must pair up (what r sends to q
    is, row for row, what q expects from r), the renumbered block must give the plain product in the STORED order, and
    `halo_sizes` (the global figure behind the `auto` choice) must equal what the plans say."""
    from geographconv_amd.dist import HaloPlan, _pattern_rows, balanced_bounds, halo_sizes
    A = None
    if symmetric:
        # a community graph numbered BY community (the generator shuffles the ids; graph.label_propagation recovers this)
        A = synth.community_ahat(1200, 9000, n_comm=6, p_in=0.9, seed=5)
        order = np.argsort(synth.community_edges(1200, 9000, 6, 0.9, seed=5)[1], kind='stable')
        A = sps.csr_matrix(A[order][:, order])
        A.sort_indices()
    else:
        A = sps.random(1200, 1200, density=0.004, format='csr', dtype=np.float32, random_state=2) + sps.identity(1200, format='csr', dtype=np.float32)
        A = sps.csr_matrix(A)
        A.sort_indices()
    N, w = A.shape[0], 5
    b = balanced_bounds(A.indptr, w, row_cost=10.0)
    b[2] = b[1]                                                          # a rank that owns NO rows
    parts = [RowPartition(N, w, r, bounds=b) for r in range(w)]
    plans = [HaloPlan(p, *_pattern_rows(A, p.r0, p.r1, symmetric)) for p in parts]
    assert np.array_equal(halo_sizes(A, b, symmetric), [pl.n_halo for pl in plans])
    Z = np.random.RandomState(0).randn(N, 5).astype(np.float32)
    At = sps.csr_matrix(A.T)
    At.sort_indices()
    for q, (p, pl) in enumerate(zip(parts, plans)):
        assert pl.recv_counts[q] == 0 and pl.send_counts[q] == 0 and pl.recv_counts.sum() == pl.n_halo
        # what arrives: rank r's packed rows for q, peer after peer
        pieces = []
        for r, (pr, plr) in enumerate(zip(parts, plans)):
            o = int(plr.send_counts[:q].sum())
            rows = plr.send_rows[o:o + int(plr.send_counts[q])]
            assert len(rows) == pl.recv_counts[r]
            pieces.append(pr.r0 + rows)
        assert np.array_equal(np.concatenate(pieces), pl.recv_cols)      # same rows, same order, no negotiation
        operand = np.concatenate([Z[p.r0:p.r1], Z[pl.recv_cols]])
        for M in (A, At):
            blk = pl.local_rows_csr(M)
            ref = M[p.r0:p.r1]
            assert np.array_equal(blk.indptr, ref.indptr) and np.array_equal(blk.data, ref.data)     # stored order kept
            assert np.array_equal(operand[blk.indices], Z[ref.indices])
    if symmetric:
        # the community graph's halo is a fraction of what an all-gather delivers; the power-law graph's is nearly all of it
        comm_frac = max(pl.n_halo / max(1, N - p.n_local) for p, pl in zip(parts, plans) if p.n_local)
        P = synth.powerlaw_ahat(1200, 9000, seed=5)
        pw = halo_sizes(P, balanced_bounds(P.indptr, w), True) / (N - np.diff(balanced_bounds(P.indptr, w)))
        assert comm_frac < pw.max()


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q, exchange):
    import torch
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from geographconv_amd import backend
        from geographconv_amd.dist import TorchDistComm
        from geographconv_amd.nn import layers as L
        from tests import cpu_backend_double
        from tests.helpers import load_case, make_clf
        backend.use(cpu_backend_double)
        out = {}
        for name in ('tiny_highway', 'tiny_plain_reg', 'tiny_odd_widths'):
            z, A, X, params, cfg = load_case(name)
            comm = TorchDistComm(cfg['N'], torch.device('cpu'), exchange=exchange)
            clf = make_clf(cfg, params, device=torch.device('cpu'), comm=comm)
            clf.inject_dropout_mask(z['mask'])
            res = []
            for step in range(2):
                o = clf.f_train(X, z['Y'][z['tr']], z['Y'][z['dev']], A, z['tr'], z['dev'])
                res.append(([float(v) for v in o[:4]], clf.gather_output(o[4]), clf.get_grads(),
                            L.get_all_param_values(clf.l_out)))
            pred, probs = clf.predict(X, A, z['te'])
            out[name] = (res, pred, probs, dict(L.DenseLayer.early_starts))
            L.DenseLayer.early_starts.update(fwd=0, bwd=0)
        if rank == 0:
            q.put(out)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("exchange,world", [("a2a", 2), ("allgather", 2), ("a2a", 3), ("a2a", 8), ("allgather", 4), ("halo", 2), ("agpipe", 2), ("agpipe", 4),
                                            ("halo", 3), ("halo", 8)])
def test_multi_rank_gloo_matches_single_process_oracle(exchange, world):
    import torch.multiprocessing as mp
    from tests.helpers import load_case
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, exchange)) for r in range(world)]
    for p in procs:
        p.start()
    out = q.get(timeout=800)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for name, (res, pred, probs, early) in out.items():
        z, A, X, params, cfg = load_case(name)
        if name == 'tiny_highway':
            # the highway blocks' convolutions started their exchange ahead of the gate layer in both sweeps
            # (2 train steps + 1 predict forward; >= 1 block)
            assert early['fwd'] >= 3 and early['bwd'] >= 2, early
        elif name == 'tiny_plain_reg':
            assert early == {'fwd': 0, 'bwd': 0}, early       # plain GCN: nothing to overlap with
        for step, (sc, P, grads, pv) in enumerate(res):
            ref = z['step%d_scalars' % step]
            assert np.allclose(sc, ref, rtol=1e-5, atol=1e-6), (name, step, sc, ref)
            assert np.allclose(P, z['step%d_P' % step], rtol=1e-4, atol=2e-6)
            for i, g in enumerate(grads):
                r = z['step%d_grad%d' % (step, i)]
                assert np.allclose(g, r, rtol=2e-4, atol=2e-7 + 1e-5 * np.abs(r).max()), (name, step, i)
            for i, q_ in enumerate(pv):
                assert np.allclose(q_, z['step%d_param%d' % (step, i)], atol=2e-5), (name, step, i)
        assert np.array_equal(pred, z['val_pred'])
        assert np.allclose(probs, z['val_probs'], rtol=1e-3, atol=5e-5)


def _asym_case():
    """Row-normalised D^-1 (A + I): NOT symmetric, so every rank must multiply by its block of the explicit transpose."""
    import scipy.sparse as sps
    from geographconv_amd import synth
    A0, X, Y = synth.small_graph(157, 7.0, 90, 9, 5, seed=4, hub=True, empty_rows=2)
    B = sps.csr_matrix((A0 != 0).astype(np.float64))
    A = sps.csr_matrix(sps.diags(1.0 / np.asarray(B.sum(1)).ravel()) @ B, dtype=np.float32)
    A.sort_indices()
    rng = np.random.RandomState(3)
    idx = rng.permutation(157)
    tr, dev, te = idx[:90].astype(np.int32), idx[90:120].astype(np.int32), idx[120:].astype(np.int32)
    cfg = dict(N=157, V=90, C=5, hid=[24, 24], highway=True, p=0.0, reg=1e-5)
    return A, X, Y, tr, dev, te, cfg


def _asym_worker(rank, world, port, q, exchange):
    import torch
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from geographconv_amd import backend
        from geographconv_amd.dist import TorchDistComm
        from tests import cpu_backend_double
        from tests.helpers import make_clf
        from oracle import gcn_oracle as O
        backend.use(cpu_backend_double)
        A, X, Y, tr, dev, te, cfg = _asym_case()
        params = O.random_params(cfg['V'], cfg['hid'], cfg['C'], True, seed=9)
        comm = TorchDistComm(cfg['N'], torch.device('cpu'), exchange=exchange)
        clf = make_clf(cfg, params, device=torch.device('cpu'), comm=comm)
        o = clf.f_train(X, Y[tr], Y[dev], A, tr, dev)
        res = ([float(v) for v in o[:4]], clf.gather_output(o[4]), clf.get_grads())     # (a collective: all ranks)
        if rank == 0:
            q.put(res)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("exchange,world", [("a2a", 3), ("allgather", 2), ("halo", 3), ("agpipe", 3)])
def test_multi_rank_asymmetric_adjacency(exchange, world):
    import torch.multiprocessing as mp
    from oracle import gcn_oracle as O
    assert abs(_asym_case()[0] - _asym_case()[0].T).max() > 1e-3
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_asym_worker, args=(r, world, port, q, exchange)) for r in range(world)]
    for p in procs:
        p.start()
    sc, P, grads = q.get(timeout=800)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    A, X, Y, tr, dev, te, cfg = _asym_case()
    params = O.random_params(cfg['V'], cfg['hid'], cfg['C'], True, seed=9)
    new, ref, rg = O.f_train(params, O.AdamState(params), X, Y[tr], Y[dev], A, tr, dev, cfg['hid'], True, 0.0, None, cfg['reg'])
    assert np.allclose(sc, [ref[0], ref[1], ref[2], ref[3]], rtol=1e-5, atol=1e-6)
    assert np.allclose(P, ref[4], rtol=1e-4, atol=2e-6)
    for i, (g, r) in enumerate(zip(grads, rg)):
        assert np.allclose(g, r, rtol=2e-4, atol=2e-7 + 1e-5 * np.abs(r).max()), i


def _random_worker(rank, world, port, q, exchange):
    """Random models through the partitioned path: N not divisible by (or smaller than) the world size, widths that do not
    divide into equal panels, tiny class counts, index sets with gaps -- every rank checks against the restatement."""
    import torch
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    err = None
    try:
        from geographconv_amd import backend
        from geographconv_amd.dist import TorchDistComm
        from geographconv_amd.nn import layers as L
        from tests import cpu_backend_double
        from tests.helpers import make_clf
        backend.use(cpu_backend_double)
        for seed in range(7):
            rng = np.random.RandomState(9000 + seed)
            N = int(rng.choice([2, 5, 23, 64, 131]))
            V = int(rng.choice([6, 19, 40]))
            C = int(rng.choice([2, 5, 17]))
            highway = bool(rng.randint(2))
            depth = int(rng.choice([1, 2, 3]))
            w = int(rng.choice([3, 8, 13]))
            hid = [w] * depth if highway else [int(rng.choice([3, 8, 13])) for _ in range(depth)]
            p = float(rng.choice([0.0, 0.4]))
            reg = float(rng.choice([0.0, 1e-3]))
            A, X, Y = synth.small_graph(N, 3.0, V, 5, C, seed=seed, empty_rows=int(rng.randint(0, 2)))
            params = O.random_params(V, hid, C, highway, seed=seed + 1, scale=0.5)
            perm = rng.permutation(N)
            n_tr = max(1, N // 2)
            tr, dv = np.sort(perm[:n_tr]).astype(np.int32), np.sort(perm[n_tr:n_tr + max(1, N // 4)]).astype(np.int32)
            if len(dv) == 0:
                dv = tr[:1].copy()
            mask = (rng.rand(N, hid[0]) < (1 - p)).astype(np.uint8) if p > 0 else np.ones((N, hid[0]), np.uint8)
            cfg = dict(N=N, V=V, C=C, hid=hid, highway=highway, p=p, reg=reg)
            comm = TorchDistComm(N, torch.device('cpu'), exchange=exchange)
            comm.prepare(A)
            clf = make_clf(cfg, [q_.copy() for q_ in params], device=torch.device('cpu'), comm=comm)
            clf.inject_dropout_mask(mask)
            st = O.AdamState(params)
            cur = [q_.copy() for q_ in params]
            for step in range(2):
                new, outs, grads = O.f_train(cur, st, X, Y[tr], Y[dv], A, tr, dv, hid, highway, p, mask.astype(np.float32), reg)
                o = clf.f_train(X, Y[tr], Y[dv], A, tr, dv)
                P = clf.gather_output(o[4])
                what = (exchange, world, seed, cfg, step)
                assert np.allclose([float(v) for v in o[:4]], outs[:4], rtol=2e-5, atol=2e-6), (what, o[:4], outs[:4])
                assert np.allclose(P, outs[4], rtol=2e-4, atol=2e-6), what
                for i, (g, r) in enumerate(zip(clf.get_grads(), grads)):
                    assert np.allclose(g, r, rtol=5e-4, atol=5e-7 + 2e-5 * np.abs(r).max()), (what, i)
                for i, (a, b) in enumerate(zip(L.get_all_param_values(clf.l_out), new)):
                    assert np.allclose(a, b, rtol=1e-4, atol=5e-5), (what, 'param', i)
                cur = new
    except BaseException as e:          # noqa: BLE001 -- reported to the parent, which fails the test
        import traceback
        err = traceback.format_exc()[-3000:]
        raise
    finally:
        if rank == 0:
            q.put(err or 'ok')
        dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("exchange,world", [("a2a", 3), ("allgather", 3), ("a2a", 4), ("halo", 4), ("agpipe", 3)])
def test_multi_rank_gloo_random_models(exchange, world):
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_random_worker, args=(r, world, port, q, exchange)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = q.get(timeout=800)
    finally:
        for p in procs:
            p.join(timeout=120)
            if p.is_alive():
                p.terminate()
    assert res == 'ok', res
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
