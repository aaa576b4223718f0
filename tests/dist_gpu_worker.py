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
    local = int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)
    dist.init_process_group('nccl', device_id=device)
    from geographconv_amd.dist import TorchDistComm
    from geographconv_amd.nn import layers as L
    from tests.helpers import load_case, make_clf
    for name, exchange in [('tiny_highway', 'a2a'), ('tiny_plain_reg', 'a2a'), ('tiny_odd_widths', 'a2a'),
                           ('tiny_highway', 'allgather'), ('tiny_odd_widths', 'allgather')]:
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
    for exchange in ('allgather', 'a2a'):
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
    for comm in (None, TorchDistComm(cfg['N'], device, exchange='a2a'), TorchDistComm(cfg['N'], device, exchange='allgather')):
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
    if dist.get_rank() == 0:
        print('DIST_GPU_OK world=%d backend=%s' % (dist.get_world_size(), type(comm.dist).__name__))
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
