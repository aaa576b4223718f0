"""Per-rank COMPUTE time of the partitioned f_train step, measured on one GPU: rank `--rank` of a simulated
`--world`-rank job runs its real share of the work while the collectives are replaced by local copies of the
same size (so packing, the narrow SpMM, the local GEMMs, ... are timed; the wire time is not).  Used to split
a multi-GPU step into compute and communication when only 1-GPU boxes are available.
   python tools/sim_rank.py --world 8 [--rank 0] [--exchange a2a|allgather] [--steps 5]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from geographconv_amd import synth  # noqa: E402
from geographconv_amd.dist import RowPartition, TorchDistComm  # noqa: E402
from geographconv_amd.gcnmodel import GraphConv  # noqa: E402


class _Done:
    def wait(self):
        return True


class FakeDist:
    capturable = True          # local copies on the launch stream: a captured step replays them

    class ReduceOp:
        SUM = 0

    def __init__(self, rank, world):
        self.rank, self.world = rank, world
        self.bytes_a2a = self.bytes_ag = self.n_a2a = self.n_ag = 0

    def get_rank(self, group=None):
        return self.rank

    def get_world_size(self, group=None):
        return self.world

    def all_to_all_single(self, out, inp, output_split_sizes=None, input_split_sizes=None, group=None, async_op=False):
        if output_split_sizes is not None:               # halo exchange: as many bytes written as a rank receives
            k = min(out.shape[0], inp.shape[0])
            out[:k].copy_(inp[:k])
            if out.shape[0] > k:
                out[k:].zero_()
            self.bytes_a2a += out.numel() * out.element_size()
            self.n_a2a += 1
            return _Done()
        out.copy_(inp)                                   # same bytes through HBM instead of xGMI
        self.bytes_a2a += inp.numel() * inp.element_size() * (self.world - 1) // self.world
        self.n_a2a += 1
        return _Done()

    def all_gather_into_tensor(self, out, inp, group=None, async_op=False):
        self.bytes_ag += out.numel() * out.element_size() * (self.world - 1) // self.world
        self.n_ag += 1
        return _Done()

    def all_reduce(self, t, op=None, group=None):
        pass


class SimComm(TorchDistComm):
    def __init__(self, N, device, rank, world, exchange):
        self.dist = FakeDist(rank, world)
        self.group = None
        self.rank, self.world = rank, world
        self.part = RowPartition(N, world, rank)
        self.device = device
        self.exchange = exchange
        self.balance = True
        self._auto = exchange == 'auto'
        if self._auto:
            self.exchange = 'a2a' if world >= 3 else 'allgather'
        self._auto_default = self.exchange
        self.halo = self.halo_rows = None
        self._bufs = {}
        from geographconv_amd import tuning
        self.slabs = max(1, int(tuning.DIST_AG_SLABS))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--world', type=int, default=8)
    ap.add_argument('--rank', type=int, default=0)
    ap.add_argument('--exchange', default='a2a')
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--shape', default='twus')
    ap.add_argument('--hid', nargs='+', type=int, default=[300, 300, 300])
    ap.add_argument('--gemm-precision', default='f32')
    ap.add_argument('--reorder', default=None, help="GraphConv(reorder=...): 'lpa' numbers communities contiguously")
    ap.add_argument('--hip-graph', action='store_true', help="capture the rank's step in a hipGraph and replay it (round 6)")
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    A, X, Y, (tr, dv, te), C = synth.make_graph(args.shape)
    comm = SimComm(A.shape[0], dev, args.rank, args.world, args.exchange)
    clf = GraphConv(X.shape[1], C, args.hid, 0.0, 0.5, highway=True, device=dev, comm=comm, gemm_precision=args.gemm_precision,
                    reorder=args.reorder, hip_graph=args.hip_graph)
    clf.build_model(A, seed=77)
    for _ in range(4 if args.hip_graph else 2):          # (two eager steps, then the capture, then a first replay)
        clf.f_train(X, Y[tr], Y[dv], A, tr, dv)
    torch.cuda.synchronize()
    d = comm.dist
    d.bytes_a2a = d.bytes_ag = d.n_a2a = d.n_ag = 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        clf.f_train(X, Y[tr], Y[dv], A, tr, dv)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / args.steps * 1e3
    if comm.halo_rows is not None:
        print('halo rows per rank and exchange: %s  (an all-gather delivers %d)' % (comm.halo_rows.tolist(), comm.part.N - comm.part.n_local))
    print('world=%d rank=%d exchange=%s %s%s rows %d (of %d): %.2f ms compute per step; per step: %d all-to-all (%.0f MB on the wire), '
          '%d all-gather (%.0f MB)' % (args.world, args.rank, comm.exchange, args.gemm_precision,
                                       ' hipGraph' if args.hip_graph and clf._hg and clf._hg.get('graph') is not None else '', comm.part.n_local, comm.part.N, ms,
                                       d.n_a2a // args.steps, d.bytes_a2a / args.steps / 1e6, d.n_ag // args.steps,
                                       d.bytes_ag / args.steps / 1e6))


if __name__ == '__main__':
    main()
