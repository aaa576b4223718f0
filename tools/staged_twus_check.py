"""Full-size check of the partitioned path on ONE GPU: `world` processes share cuda:0 (collectives staged through the host
and gloo, as tests/dist_gpu_worker.py does), every rank runs its real share of two TwitterUS-shape training steps, rank 0
then runs the same two steps un-partitioned and compares losses, hit counts and the gathered probabilities.
   python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tools/staged_twus_check.py [a2a|allgather] [cfg5]
(`cfg5`: BASELINE configs[4] -- six 600-wide highway layers in the bf16 configuration -- with its own tolerances)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402



def main():
    exchange = sys.argv[1] if len(sys.argv) > 1 else 'a2a'
    cfg5 = len(sys.argv) > 2 and sys.argv[2] == 'cfg5'
    hid = [600] * 6 if cfg5 else [300, 300, 300]
    prec = 'bf16' if cfg5 else None
    os.environ['GEOGCN_DIST_BACKEND'] = 'staged-gloo'
    from geographconv_amd import dist as gdist
    device = gdist.init_process_group(0)
    from geographconv_amd import synth
    from geographconv_amd.dist import TorchDistComm
    from geographconv_amd.gcnmodel import GraphConv
    A, X, Y, (tr, dv, te), C = synth.make_graph('twus')
    N = A.shape[0]
    mask = (np.random.RandomState(3).rand(N, hid[0]) < 0.5).astype(np.uint8)
    comm = TorchDistComm(N, device, exchange=exchange)

    def run(c):
        clf = GraphConv(X.shape[1], C, hid, 0.0, 0.5, highway=True, device=device, comm=c, gemm_precision=prec)
        clf.build_model(A, seed=77)
        clf.inject_dropout_mask(mask)
        out = []
        for _ in range(2):
            o = clf.f_train(X, Y[tr], Y[dv], A, tr, dv)
            out.append(([float(v) for v in o[:4]], clf.gather_output(o[4])))
        return out

    got = run(comm)
    dist.barrier()
    if dist.get_rank() == 0:
        want = run(None)
        for step, (g, w) in enumerate(zip(got, want)):
            print('step %d  partitioned %s   one GPU %s' % (step, g[0], w[0]), flush=True)
            assert np.allclose(g[0], w[0], rtol=2e-3 if cfg5 else 2e-5, atol=2e-4 if cfg5 else 2e-6), (g[0], w[0])
            d = np.abs(g[1] - w[1]).max()
            same = float((g[1].argmax(1) == w[1].argmax(1)).mean())
            print('        max |dP| %.3g   argmax agreement %.6f' % (d, same), flush=True)
            assert d < (2e-3 if cfg5 else 2e-5) and same > (0.995 if cfg5 else 0.9999)
        print('STAGED_TWUS_OK world=%d exchange=%s %s' % (dist.get_world_size(), exchange, 'configs[4]: 6x600 bf16' if cfg5 else '3x300 f32'), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
