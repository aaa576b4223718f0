#!/usr/bin/env python
"""Headline benchmark (driver contract): GCN-layer fwd+bwd edges/s on the TwitterUS-shape graph.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one full-graph ``GraphConv.f_train``: forward + backward + Adam of the 3x300 highway
GCN (BASELINE.json configs[2]: synthetic TwitterUS-shape CSR, N=440,000, nnz(A_hat)=10,730,596,
V=10,000, C=256, fp32, dropout 0.5) with (A_hat, X, Y) already resident in HBM.  Every step runs
3 graph-convolution layers forward and backward over all stored edges, so

    value = n_conv_layers * nnz(A_hat) * K / t_K        [GCN-layer fwd+bwd edges/s, whole job]

With --gpus N the SAME graph is row-partitioned over N ranks (strong scaling; per conv layer and
direction the SpMM operand is exchanged over RCCL: in-place all-gather at 2 ranks, a feature
repartition with two all-to-alls from 3 ranks -- DESIGN.md section 5).  The JSON line also carries
  roofline     : the dominant kernel (spmm_rows_kernel, A_hat.Z at F=300), algorithmic bytes / its
                 average launch duration measured live with hipEvents on the launch stream;
  cpu_baseline : the NumPy/SciPy oracle (oracle/gcn_oracle.py, kind "port") timed on this box's
                 host cores on a bounded sample (rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def spmm_algorithmic_bytes(n_rows_out, n_cols, nnz, F, b_bytes=4):
    """every operand touched exactly once (fp32 values, int32 indices); the gathered operand B is
    fp32, or bf16 (b_bytes = 2) in the bf16 configuration."""
    return 8 * nnz + 4 * (n_rows_out + 1) + b_bytes * n_cols * F + 4 * n_rows_out * F


def cpu_baseline(shape, A, X, Y, tr, dev, hid, C, sample):
    """Oracle timed on the host: one f_train step ('step') or one conv layer fwd+bwd ('layer')."""
    from oracle import gcn_oracle as O
    nnz = A.nnz
    threads = os.cpu_count() or 1
    model = ''
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                model = line.split(':', 1)[1].strip()
                break
    except OSError:
        pass
    if sample == 'layer':
        rng = np.random.RandomState(1)
        H = rng.randn(A.shape[0], 300).astype(np.float32)
        G = rng.randn(A.shape[0], 300).astype(np.float32)
        W = (rng.randn(300, 300) * 0.05).astype(np.float32)
        b = np.zeros(300, np.float32)
        t0 = time.time()
        O.conv_layer_fwd_bwd(H, W, b, A, G)
        t = time.time() - t0
        return {"value": nnz / t, "unit": "edges/s", "cores": threads, "cpu_model": model, "kind": "port", "seconds": round(t, 2),
                "sample": "1 ConvolutionDenseLayer2 fwd+bwd (300->300) on the full %s graph; scipy CSR SpMM is "
                          "single-threaded like Theano's StructuredDot, BLAS sgemm uses %d threads" % (shape, threads)}
    params = O.random_params(X.shape[1], hid, C, True, seed=7)
    mask = (np.random.RandomState(3).rand(X.shape[0], hid[0]) < 0.5).astype(np.float32)
    st = O.AdamState(params)
    t0 = time.time()
    O.f_train(params, st, X, Y[tr], Y[dev], A, tr, dev, hid, True, 0.5, mask)
    t = time.time() - t0
    n_conv = len(hid)
    return {"value": n_conv * nnz / t, "unit": "edges/s", "cores": threads, "cpu_model": model, "kind": "port", "seconds": round(t, 2),
            "sample": "1 full f_train step (same workload, same unit: %d conv layers x nnz / step time) on the full %s "
                      "graph; scipy CSR SpMM is single-threaded like Theano's StructuredDot, BLAS sgemm uses %d "
                      "threads" % (n_conv, shape, threads)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--shape', default='twus', choices=['twus', 'cmu'])
    ap.add_argument('--hid', nargs='+', type=int, default=[300, 300, 300])
    ap.add_argument('--dropout', type=float, default=0.5)
    ap.add_argument('--gemm-precision', default='f32', choices=['f32', 'bf16x3', 'bf16'],
                    help='f32 = exact fp32 MFMA (the headline configuration)')
    ap.add_argument('--cpu-sample', default='step', choices=['step', 'layer', 'none'])
    args = ap.parse_args()

    import torch
    from geographconv_amd import ops, synth
    from geographconv_amd.gcnmodel import GraphConv

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs one process per GPU: launch with python -m torch.distributed.run "
                             "--nproc-per-node %d bench.py --gpus %d" % (args.gpus, args.gpus, args.gpus))
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
    ops.require_gpu()
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)
    comm = None
    force_dist = os.environ.get('GEOGCN_BENCH_FORCE_DIST') == '1'      # exercise the partitioned path at world 1
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        dist.init_process_group('nccl', device_id=device)

    t0 = time.time()
    A, X, Y, (tr, dev, te), C = synth.make_graph(args.shape)
    N, nnz = A.shape[0], int(A.nnz)
    if rank == 0:
        log('[bench] %s graph generated in %.1fs: N=%d nnz(A)=%d nnz(X)=%d' % (args.shape, time.time() - t0, N, nnz, X.nnz))
    if world > 1 or force_dist:
        from geographconv_amd.dist import TorchDistComm
        comm = TorchDistComm(N, device)

    clf = GraphConv(X.shape[1], C, args.hid, 0.0, args.dropout, highway=True, device=device, comm=comm,
                    gemm_precision=args.gemm_precision)
    clf.build_model(A, seed=77)
    clf._force_dist = force_dist
    y_tr, y_dev = Y[tr], Y[dev]

    def step():
        return clf.f_train(X, y_tr, y_dev, A, tr, dev)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    # roofline leg: library-side hipEvent pairs around the F=hid SpMM row kernel, on the launch stream
    timer = ops.SpmmTimer(capacity=max(16, 8 * args.steps))
    # the dense operand of the timed SpMM: F = hid at one GPU; one feature panel per rank under the a2a scheme
    F_spmm = args.hid[-1]
    if comm is not None and comm.exchange == 'a2a':
        F_spmm = comm.panel_width(args.hid[-1])
    timer.attach(only_F=F_spmm, only_nnz=clf._device_graph(X, A)['A'].fwd.nnz)
    barrier()
    t0 = time.perf_counter()
    last = None
    for _ in range(args.steps):
        last = step()
    barrier()
    t = time.perf_counter() - t0
    timer.detach()
    kern_ms = timer.read_ms()

    if world > 1:
        tt = torch.tensor([t], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        t = float(tt.item())
    n_conv = len(args.hid)
    value = n_conv * nnz * args.steps / t

    if rank == 0:
        g = clf._device_graph(X, A)
        csr = g['A'].fwd
        rp = csr.rowptr_host
        deg = np.diff(rp)
        F = F_spmm
        # one launch of spmm_rows_kernel covers every stored edge of the local row block (short rows with
        # the fused epilogue + the 128-nonzero chunks of the long rows): algorithmic bytes = SURVEY.md §8d
        bf16_operand = args.gemm_precision == 'bf16' and world == 1       # (the partitioned path exchanges fp32)
        alg = spmm_algorithmic_bytes(len(deg), csr.shape[1], int(csr.nnz), F, 2 if bf16_operand else 4)
        e_short = int(csr.nnz)
        avg_ms = float(np.mean(kern_ms)) if kern_ms else None
        achieved = alg / (avg_ms * 1e-3) / 1e9 if kern_ms else None
        traffic = None            # PMC passes are collected at 1 GPU, F = hid (profiles/pmc_spmm_latest.json)
        pmc_file = os.path.join(ROOT, 'profiles', 'pmc_spmm_bf16_latest.json' if bf16_operand else 'pmc_spmm_latest.json')
        if os.path.exists(pmc_file) and world == 1 and F == 300 and args.shape == 'twus':
            try:
                traffic = json.load(open(pmc_file)).get('hbm_bytes_per_launch')
            except Exception:
                traffic = None
        roofline = {"bound": "hbm", "kernel": "spmm_rows_kernel<%d,*> (A_hat.Z, F=%d%s)" % (
                        (F + 63) // 64, F, "" if world == 1 else ", rank 0's feature panel of all rows"),
                    "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": (achieved / HBM_PEAK_GBPS) if achieved else None, "traffic": traffic,
                    "algorithmic_bytes_per_launch": alg, "avg_launch_ms": avg_ms, "launches_timed": len(kern_ms),
                    "edges_per_launch": e_short, "bytes_per_edge": alg / max(1, e_short)}
        out = {
            "metric": "GCN-layer fwd+bwd edges/sec", "value": value, "unit": "edges/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": t / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32" if args.gemm_precision != "bf16" else "bf16", "data": "synthetic",
            "config": {"workload": "TwitterUS-shape synthetic power-law CSR (BASELINE configs[2]): N=%d, nnz(A_hat)=%d, "
                                   "X %dx%d nnz=%d, C=%d; %s highway GCN, dropout %.2f, Adam; full-graph f_train step"
                                   % (N, nnz, N, X.shape[1], X.nnz, C, 'x'.join(map(str, args.hid)), args.dropout),
                       "edges_per_step": n_conv * nnz, "parallelism": "rows%d" % world if world > 1 else "single",
                       "gemm": {"f32": "exact fp32 MFMA (v_mfma_f32_16x16x4_f32)", "bf16x3": "3-term bf16 split MFMA, fp32 accumulate",
                                "bf16": "bf16 MFMA, fp32 accumulate"}[args.gemm_precision],
                       "train_loss_last": float(last[0])},
            "roofline": roofline,
        }
        if world == 1 and args.cpu_sample != 'none':
            log('[bench] timing the CPU oracle (%s sample)...' % args.cpu_sample)
            out["cpu_baseline"] = cpu_baseline(args.shape, A, X, Y, tr, dev, args.hid, C, args.cpu_sample)
        else:
            out["cpu_baseline"] = None
        # RCCL prints a version banner through C stdio; flush it so that the JSON line is the LAST line of stdout
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(out), flush=True)
    if world > 1 or force_dist:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
