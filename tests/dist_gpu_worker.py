"""Worker for tests/test_dist_gpu.py: runs under torch.distributed.run with the nccl (RCCL) backend and
drives GraphConv through TorchDistComm -- the exact code path `bench.py --gpus N` uses -- on however many
GPUs the launcher gave it, comparing against the committed golden vectors."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    from geographconv_amd import dist as gdist
    device = gdist.init_process_group(int(os.environ.get('LOCAL_RANK', '0')))      # nccl, or staged-gloo on cuda:0
    from geographconv_amd.dist import TorchDistComm
    from geographconv_amd.nn import layers as L
    from tests.helpers import load_case, make_clf
    for name, exchange in [('tiny_highway', 'a2a'), ('tiny_plain_reg', 'a2a'), ('tiny_odd_widths', 'a2a'),
                           ('tiny_highway', 'allgather'), ('tiny_odd_widths', 'allgather'), ('tiny_highway', 'halo'),
                           ('tiny_highway', 'agpipe'), ('tiny_plain_reg', 'agpipe'), ('tiny_odd_widths', 'agpipe'),
                           ('tiny_plain_reg', 'halo'), ('tiny_odd_widths', 'halo')]:
        z, A, X, params, cfg = load_case(name)
        comm = TorchDistComm(cfg['N'], device, exchange=exchange)
        clf = make_clf(cfg, params, device=device, comm=comm)
        clf.inject_dropout_mask(z['mask'])
        # force the distributed branches even at world_size 1
        clf._force_dist = True
        for step in range(2):
            o = clf.f_train(X, z['Y'][z['tr']], z['Y'][z['dev']], A, z['tr'], z['dev'])
            ref = z['step%d_scalars' % step]
            assert np.allclose([float(v) for v in o[:4]], ref, rtol=1e-5, atol=1e-6), (name, step)
            with np.testing.assert_raises(RuntimeError):
                np.asarray(o[4])                    # sharded: no collective hidden in a conversion
            P = clf.gather_output(o[4])
            assert np.allclose(P, z['step%d_P' % step], rtol=1e-4, atol=2e-6), (name, step)
            for i, g in enumerate(clf.get_grads()):
                r = z['step%d_grad%d' % (step, i)]
                assert np.allclose(g, r, rtol=2e-4, atol=2e-7 + 1e-5 * np.abs(r).max()), (name, step, i)
        pred, probs = clf.predict(X, A, z['te'])
        srt = np.sort(z['val_probs'], axis=1)
        safe = (srt[:, -1] - srt[:, -2]) > 1e-4
        assert np.array_equal(pred[safe], z['val_pred'][safe]), name
    # the partitioned graph product against the one-GPU kernel on the same rows: BITWISE (row order and per-row
    # accumulation order do not depend on the partition; meaningful from 2 ranks on, trivially true at 1)
    from geographconv_amd import ops, synth
    A = synth.powerlaw_ahat(6000, 70000, seed=1)
    Zh = np.random.RandomState(1).randn(6000, 44).astype(np.float32)
    bias = torch.from_numpy(np.random.RandomState(2).randn(44).astype(np.float32)).to(device)
    full = ops.spmm(ops.CSR(A, device), ops.DMat.from_numpy(Zh, device), bias=bias, act=ops.ACT_TANH)
    for exchange in ('allgather', 'a2a', 'halo', 'agpipe'):
        comm = TorchDistComm(6000, device, exchange=exchange)
        comm.prepare(A)
        dA = comm.graph_operand(A)
        z = comm.matmul_target(44, tag='t', direct=False)
        z.copy_from(ops.DMat.from_numpy(Zh[comm.part.r0:comm.part.r1], device))
        out = comm.graph_spmm(dA.fwd, z, bias, ops.ACT_TANH, 44, tag='t')
        assert torch.equal(out.t[:, :44], full.t[comm.part.r0:comm.part.r1, :44]), exchange
    # bf16 configuration through the partitioned path (bf16 operand on the wire): agrees with the one-GPU bf16 run
    z_, A_, X_, params, cfg = load_case('tiny_highway')
    ref = None
    for comm in (None, TorchDistComm(cfg['N'], device, exchange='a2a'), TorchDistComm(cfg['N'], device, exchange='allgather'),
                 TorchDistComm(cfg['N'], device, exchange='halo'), TorchDistComm(cfg['N'], device, exchange='agpipe')):
        from geographconv_amd.gcnmodel import GraphConv
        clf = GraphConv(cfg['V'], cfg['C'], cfg['hid'], cfg['reg'], cfg['p'], highway=True, device=device, comm=comm,
                        gemm_precision='bf16')
        clf.build_model(None, seed=77)
        L.set_all_param_values(clf.l_out, params)
        clf.inject_dropout_mask(z_['mask'])
        clf._force_dist = comm is not None
        o = clf.f_train(X_, z_['Y'][z_['tr']], z_['Y'][z_['dev']], A_, z_['tr'], z_['dev'])
        got = ([float(v) for v in o[:4]], clf.gather_output(o[4]), clf.get_grads())
        if ref is None:
            ref = got
        else:
            assert np.allclose(got[0], ref[0], rtol=2e-3, atol=1e-4), (got[0], ref[0])
            assert np.allclose(got[1], ref[1], rtol=2e-2, atol=2e-3)
            for g, r in zip(got[2], ref[2]):
                assert np.allclose(g, r, rtol=5e-2, atol=5e-3 * np.abs(r).max() + 1e-7)
    # a mid-size model (rows, widths and classes that do not divide by the world size; hub rows -> chunked long rows in
    # every rank's block): partitioned vs the same process's one-GPU run
    from geographconv_amd import synth as _synth
    from oracle import gcn_oracle as O
    A_, X_, Y_ = _synth.small_graph(5003, 9.0, 700, 12, 11, seed=5, empty_rows=3)
    hid, C_ = [52, 52], 11
    params = O.random_params(700, hid, C_, True, seed=6, scale=0.3)
    rng = np.random.RandomState(7)
    perm = rng.permutation(5003)
    tr_, dv_ = np.sort(perm[:2500]).astype(np.int32), np.sort(perm[2500:3500]).astype(np.int32)
    mask_ = (rng.rand(5003, 52) < 0.5).astype(np.uint8)
    ref = None
    for exchange in (None, 'a2a', 'allgather', 'halo', 'agpipe'):
        comm = None if exchange is None else TorchDistComm(5003, device, exchange=exchange)
        clf = GraphConv(700, C_, hid, 0.0, 0.5, highway=True, device=device, comm=comm)
        clf.build_model(None, seed=77)
        L.set_all_param_values(clf.l_out, [q.copy() for q in params])
        clf.inject_dropout_mask(mask_)
        res = []
        for step in range(2):
            o = clf.f_train(X_, Y_[tr_], Y_[dv_], A_, tr_, dv_)
            res.append(([float(v) for v in o[:4]], clf.gather_output(o[4]), clf.get_grads()))
        if ref is None:
            ref = res
            continue
        for step, (got, want) in enumerate(zip(res, ref)):
            assert np.allclose(got[0], want[0], rtol=2e-5, atol=2e-6), (exchange, step, got[0], want[0])
            assert np.allclose(got[1], want[1], rtol=2e-4, atol=2e-6), (exchange, step)
            for i, (g, r) in enumerate(zip(got[2], want[2])):
                assert np.allclose(g, r, rtol=5e-4, atol=5e-7 + 2e-5 * np.abs(r).max()), (exchange, step, i)
    # a graph WITH locality (communities numbered contiguously): `auto` must pick the halo exchange from 2 ranks on, receive a
    # fraction of what an all-gather delivers, and reproduce the one-GPU run
    import scipy.sparse as sps
    Nc, ncomm = 6000, 12
    A_ = _synth.community_ahat(Nc, 70000, n_comm=ncomm, p_in=0.9, seed=11)
    order = np.argsort(_synth.community_edges(Nc, 70000, ncomm, 0.9, seed=11)[1], kind='stable')
    A_ = sps.csr_matrix(A_[order][:, order])
    A_.sort_indices()
    X_ = _synth.bow_x(Nc, 500, 10, seed=12)
    Y_ = _synth.labels(Nc, 9, seed=13)
    hid, C_ = [36, 36], 9
    params = O.random_params(500, hid, C_, True, seed=14, scale=0.3)
    perm = np.random.RandomState(15).permutation(Nc)
    tr_, dv_ = np.sort(perm[:3000]).astype(np.int32), np.sort(perm[3000:4000]).astype(np.int32)
    ref = None
    for auto in (False, True):
        comm = TorchDistComm(Nc, device) if auto else None
        clf = GraphConv(500, C_, hid, 0.0, 0.0, highway=True, device=device, comm=comm)
        clf.build_model(None, seed=77)
        L.set_all_param_values(clf.l_out, [q.copy() for q in params])
        res = []
        for step in range(2):
            o = clf.f_train(X_, Y_[tr_], Y_[dv_], A_, tr_, dv_)
            res.append(([float(v) for v in o[:4]], clf.gather_output(o[4]), clf.get_grads()))
        if not auto:
            ref = res
            continue
        if dist.get_world_size() > 1:
            assert comm.exchange == 'halo', comm.exchange
            remote = Nc - np.diff(comm.part.bounds_all)
            assert comm.halo.n_halo == comm.halo_rows[comm.rank] and comm.halo_rows.max() <= 0.5 * remote.min(), comm.halo_rows
        for step, (got, want) in enumerate(zip(res, ref)):
            assert np.allclose(got[0], want[0], rtol=2e-5, atol=2e-6), ('auto-halo', step, got[0], want[0])
            assert np.allclose(got[1], want[1], rtol=2e-4, atol=2e-6), ('auto-halo', step)
            for i, (g, r) in enumerate(zip(got[2], want[2])):
                assert np.allclose(g, r, rtol=5e-4, atol=5e-7 + 2e-5 * np.abs(r).max()), ('auto-halo', step, i)
    # random small models (sizes smaller than / not divisible by the world size, odd widths, tiny class counts) against
    # the CPU restatement -- the same generator as tests/test_dist_cpu.py, here with the HIP kernels
    for seed in range(int(os.environ.get('GEOGCN_TEST_DIST_SEEDS', '8'))):
        rng = np.random.RandomState(9000 + seed)
        N = int(rng.choice([2, 5, 23, 64, 131]))
        V = int(rng.choice([6, 19, 40]))
        Cn = int(rng.choice([2, 5, 17]))
        highway = bool(rng.randint(2))
        depth = int(rng.choice([1, 2, 3]))
        w = int(rng.choice([3, 8, 13]))
        hid = [w] * depth if highway else [int(rng.choice([3, 8, 13])) for _ in range(depth)]
        p = float(rng.choice([0.0, 0.4]))
        reg = float(rng.choice([0.0, 1e-3]))
        A_, X_, Y_ = _synth.small_graph(N, 3.0, V, 5, Cn, seed=seed, empty_rows=int(rng.randint(0, 2)))
        params = O.random_params(V, hid, Cn, highway, seed=seed + 1, scale=0.5)
        perm = rng.permutation(N)
        n_tr = max(1, N // 2)
        tr_, dv_ = np.sort(perm[:n_tr]).astype(np.int32), np.sort(perm[n_tr:n_tr + max(1, N // 4)]).astype(np.int32)
        if len(dv_) == 0:
            dv_ = tr_[:1].copy()
        mask_ = (rng.rand(N, hid[0]) < (1 - p)).astype(np.uint8) if p > 0 else np.ones((N, hid[0]), np.uint8)
        for exchange in ('a2a', 'allgather', 'halo', 'agpipe'):
            comm = TorchDistComm(N, device, exchange=exchange)
            comm.prepare(A_)
            # (every third model also goes through a node reordering: invisible to the caller, partitioned or not)
            ro_kw = {'reorder': ('degree', 'rcm', 'lpa')[(seed // 3) % 3]} if seed % 3 == 2 and N > 8 else {}
            clf = GraphConv(V, Cn, hid, reg, p, highway=highway, device=device, comm=comm, **ro_kw)
            clf.build_model(A_ if ro_kw else None, seed=77)
            L.set_all_param_values(clf.l_out, [q.copy() for q in params])
            clf.inject_dropout_mask(mask_)
            clf._force_dist = True
            st = O.AdamState(params)
            cur = [q.copy() for q in params]
            for step in range(2):
                new, outs, grads = O.f_train(cur, st, X_, Y_[tr_], Y_[dv_], A_, tr_, dv_, hid, highway, p, mask_.astype(np.float32), reg)
                o = clf.f_train(X_, Y_[tr_], Y_[dv_], A_, tr_, dv_)
                what = (exchange, seed, N, V, Cn, hid, highway, p, reg, step)
                assert np.allclose([float(v) for v in o[:4]], outs[:4], rtol=2e-5, atol=2e-6), (what, o[:4], outs[:4])
                assert np.allclose(clf.gather_output(o[4]), outs[4], rtol=2e-4, atol=2e-6), what
                for i, (g, r) in enumerate(zip(clf.get_grads(), grads)):
                    assert np.allclose(g, r, rtol=5e-4, atol=5e-7 + 2e-5 * np.abs(r).max()), (what, i)
                cur = new
        # the bf16 configuration (bf16 operand and wire) on the same model: partitioned vs this process's one-GPU bf16 run
        if seed % 2 == 0:
            ref = None
            for exchange in (None, 'a2a', 'allgather', 'halo', 'agpipe'):
                comm = None if exchange is None else TorchDistComm(N, device, exchange=exchange)
                if comm is not None:
                    comm.prepare(A_)
                clf = GraphConv(V, Cn, hid, reg, p, highway=highway, device=device, comm=comm, gemm_precision='bf16')
                clf.build_model(None, seed=77)
                L.set_all_param_values(clf.l_out, [q.copy() for q in params])
                clf.inject_dropout_mask(mask_)
                clf._force_dist = comm is not None
                o = clf.f_train(X_, Y_[tr_], Y_[dv_], A_, tr_, dv_)
                got = ([float(v) for v in o[:4]], clf.gather_output(o[4]), clf.get_grads())
                if ref is None:
                    ref = got
                    continue
                what = ('bf16', exchange, seed, N, V, Cn, hid, highway, p)
                assert np.allclose(got[0], ref[0], rtol=2e-3, atol=2e-4), (what, got[0], ref[0])
                assert np.allclose(got[1], ref[1], rtol=2e-2, atol=2e-3), what
                for i, (g, r) in enumerate(zip(got[2], ref[2])):
                    assert np.allclose(g, r, rtol=5e-2, atol=5e-3 * np.abs(r).max() + 1e-6), (what, i)
    # (round 6) the partitioned step captured in a hipGraph and replayed (GraphConv(hip_graph=True); transports that are stream-ordered
    # RCCL only -- comm.capturable): six steps with the Philox dropout stream follow the eager run of the same partitioned model
    A_, X_, Y_ = _synth.small_graph(5003, 9.0, 700, 12, 11, seed=5, empty_rows=3)
    hid, C_ = [52, 52], 11
    params = O.random_params(700, hid, C_, True, seed=6, scale=0.3)
    perm = np.random.RandomState(7).permutation(5003)
    tr_, dv_ = np.sort(perm[:2500]).astype(np.int32), np.sort(perm[2500:3500]).astype(np.int32)
    n_captured = 0
    # (host-staged transports -- the multi-rank one-GPU runs -- cannot be captured: GraphConv then stays eager, nothing to compare)
    # all-to-all based schemes (a2a, halo) are not capturable -- comm.capturable says so (tools/rccl_capture_probe.py) -- and stay eager:
    # asking for hip_graph=True with them must simply run eager steps
    for exchange in ('allgather', 'agpipe', 'a2a') if TorchDistComm(5003, device, exchange='allgather').capturable else ():
        runs = {}
        for mode in (False, True):
            comm = TorchDistComm(5003, device, exchange=exchange)
            clf = GraphConv(700, C_, hid, 1e-6, 0.5, highway=True, device=device, comm=comm, hip_graph=mode)
            clf.build_model(None, seed=77)
            L.set_all_param_values(clf.l_out, [q.copy() for q in params])
            clf._force_dist = True
            hist = []
            for step in range(6):
                o = clf.f_train(X_, Y_[tr_], Y_[dv_], A_, tr_, dv_)
                hist.append([float(v) for v in o[:4]])
            runs[mode] = (hist, clf.gather_output(o[4]), L.get_all_param_values(clf.l_out), clf)
        (he, Pe, pe, _), (hg, Pg, pg, clfg) = runs[False], runs[True]
        captured = clfg._hg is not None and clfg._hg.get('graph') is not None
        assert captured == bool(comm.capturable), (exchange, captured, comm.capturable)
        n_captured += int(captured)
        assert len(set(h[0] for h in hg)) == 6, exchange                        # six different steps, not one replayed
        for a, b in zip(he, hg):
            assert abs(a[0] - b[0]) <= 1e-5 * abs(a[0]) and abs(a[2] - b[2]) <= 1e-5 * abs(a[2]), (exchange, a, b)
        assert np.abs(Pe - Pg).max() <= 1e-5, exchange
        for q, r in zip(pe, pg):
            assert np.abs(q - r).max() <= 2e-3 * 0.05 + 1e-7 and np.mean(np.abs(q - r)) <= 1e-7, exchange
    if dist.get_rank() == 0:
        print('DIST_GPU_OK world=%d backend=%s captured_schemes=%d' % (dist.get_world_size(), type(comm.dist).__name__, n_captured))
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
