#!/usr/bin/env python
"""Headline benchmark (driver contract): GCN-layer fwd+bwd edges/s on the TwitterUS-shape graph.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one full-graph ``GraphConv.f_train``: forward + backward + Adam of the 3x300 highway
GCN (BASELINE.json configs[2]: synthetic TwitterUS-shape CSR, N=440,000, nnz(A_hat)=10,730,596,
V=10,000, C=256, fp32 in HBM and in every accumulator, dropout 0.5) with (A_hat, X, Y) already resident in HBM.  Every step runs
3 graph-convolution layers forward and backward over the stored edges, so

    value = n_conv_layers * nnz(A_hat) * K / t_K        [GCN-layer fwd+bwd edges/s, whole job]

(nominal: the output layer's backward product only walks the edges into the training nodes, because the
cross-entropy gradient is zero elsewhere; `config.edges_traversed_per_step` has the exact count.)

GEMM precision: `geographconv_amd.tuning.GEMM_PRECISION` (default "bf16x3": fp32-class split-bf16 products where a kernel takes the shape,
exact fp32 elsewhere; `dtype` stays "f32" -- inputs, outputs and accumulation are fp32 -- and `config.gemm` says what ran); at N = 1 the
same job is timed again with the exact fp32 MFMA in every product (`alt_exact_f32`).

With --gpus N the SAME graph is row-partitioned over N ranks (strong scaling; per conv layer and direction the SpMM operand is exchanged
over RCCL -- DESIGN.md section 5).  The north_star's scheme (1-D row split of A_hat + all-gather of H) is timed FIRST and its line is
printed at once (marked "early line"); the other schemes (feature repartition with two all-to-alls, slab-pipelined all-gather) and the
partition check follow, each behind a try/except and a wall-clock guard (--scheme-timeout): whatever happens to them, the last line of
stdout is one JSON line whose `value` is the fastest scheme that finished (`exchange_choice`, `alt`, `alt2`; a failed scheme is reported
as {"exchange": ..., "error": ...}); `config.dist` carries the world size seen, the RCCL version and every rank's device.  Round 6: the
--graph: the fastest capturable scheme (the all-gather ones: RCCL's all-to-all is not safe under capture, tools/rccl_capture_probe.py) is
timed once more with its step captured in a hipGraph and replayed ("<scheme>+hipgraph", `config.hip_graph`; a candidate for `value` only
if its last training loss follows the eager run's; off by default; never under the host-staged backend).
The JSON line also carries
  roofline       : the dominant kernel (spmm_rows_kernel, A_hat^T . dS at F=300: the two plain full-graph products of
                   a step), algorithmic bytes / average duration measured live with library-side hipEvent pairs on the
                   launch stream around every such product inside the timed region; `traffic` = the kernel's HBM-side bytes per
                   launch counted by THIS invocation at N = 1 (--traffic live, the default: two `rocprofv3 --pmc` child passes of the same
                   job for 2 steps, FETCH_SIZE and WRITE_SIZE separately, after the timed region; a stored pass under profiles/ when the
                   counters cannot run -- `traffic_source` says which); `gather_ceiling_tbps` = the rate at which this GPU
                   gathers rows of the product's size from a table beyond L2, measured by this invocation (tools/micro/gather_bw.hip),
                   and `frac_of_gather_ceiling` = counted traffic / launch time / that ceiling; `others` = the other hot kernels timed
                   inside further steps right after the timed region;
  layer          : SURVEY.md 8d (ii): one ConvolutionDenseLayer2 forward + backward and one highway block on the full graph, edges/s;
  step_ms        : median / p10 / p90 of the per-step times (events per step, same region);
  cpu_baseline   : the NumPy/SciPy oracle (oracle/gcn_oracle.py, kind "port": SpMM single-threaded like Theano's
                   StructuredDot, BLAS sgemm on all threads) timed on this box's host cores (rank 0, N=1 only);
  cpu_baseline_mt: the same step with EVERY pass on all cores (oracle/cpu_mt.py::f_train: sparse products and the fused
                   elementwise / softmax / column-sum passes in OpenMP, dense products in BLAS) -- the "fair multi-core CPU";
  clocks         : shader / memory clock, socket power and temperature sampled while further steps run after the timed region.
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
MFMA_F32_PEAK_TFLOPS = 157.3    # v_mfma_f32_16x16x4_f32, dense (MI355X_MICROARCH.md)
MFMA_BF16_PEAK_TFLOPS = 2500.0  # v_mfma_f32_*_bf16, dense (MI355X_MICROARCH.md: ~2.5 PF; 2:1 sparsity figures are not a peak)


class ClockSampler:
    """SURVEY.md section 8d "clocks / power state logged": a thread that reads the amdgpu hwmon files of the device (shader clock
    freq1_input, memory clock freq2_input, socket power power1_input, temperature) every ~15 ms while `n` extra training steps
    run AFTER the timed region (the same workload; never inside the region, so that the sampler cannot perturb `value`)."""

    def __init__(self, index=0):
        import glob
        self.files = {}
        cards = sorted(glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*'))
        cards = [c for c in cards if os.path.exists(os.path.join(c, 'freq1_input'))]
        self.card = None
        if cards:
            hw = cards[min(index, len(cards) - 1)]
            self.card = "hwmon #%d of %d (no PCI match)" % (min(index, len(cards) - 1), len(cards))
            # A box can show the hwmon directories of GPUs this process cannot see: take the one whose PCI address is the device's.
            try:
                import torch
                pr = torch.cuda.get_device_properties(index)
                want = "%04x:%02x:%02x." % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
                for c in cards:
                    pci = os.path.basename(os.path.realpath(os.path.join(c, '..', '..')))
                    if pci.startswith(want):
                        hw, self.card = c, "PCI " + pci
                        break
            except (AttributeError, RuntimeError, AssertionError):
                pass
            for key, name in (('sclk_mhz', 'freq1_input'), ('mclk_mhz', 'freq2_input'), ('power_w', 'power1_input'),
                              ('power_cap_w', 'power1_cap'), ('temp_c', 'temp2_input')):
                f = os.path.join(hw, name)
                if os.path.exists(f):
                    self.files[key] = f
            self.level = os.path.join(os.path.dirname(os.path.dirname(hw)), 'power_dpm_force_performance_level')
        self.samples = {k: [] for k in self.files}
        self._stop = False

    def _read(self, f):
        try:
            return float(open(f).read().strip())
        except (OSError, ValueError):
            return None

    def _loop(self):
        scale = {'sclk_mhz': 1e-6, 'mclk_mhz': 1e-6, 'power_w': 1e-6, 'power_cap_w': 1e-6, 'temp_c': 1e-3}
        while not self._stop:
            for k, f in self.files.items():
                v = self._read(f)
                if v is not None:
                    self.samples[k].append(v * scale[k])
            time.sleep(0.015)

    def run(self, step, seconds, sync):
        """Steps for `seconds` of wall time (the SMU's power / clock telemetry is a moving average over roughly a second: ten steps
        would report the idle state before them), sampling throughout; the figures of the LAST second are reported."""
        if not self.files:
            return {"error": "no amdgpu hwmon files on this box"}
        import threading
        th = threading.Thread(target=self._loop, daemon=True)
        sync()
        th.start()
        t0, n = time.perf_counter(), 0
        while time.perf_counter() - t0 < seconds:
            step()
            n += 1
            if n % 8 == 0:
                sync()               # (keeps the launch queue short, so that wall time tracks device time)
        sync()
        self._stop = True
        th.join()
        wall = time.perf_counter() - t0
        keep = max(1, int(len(next(iter(self.samples.values()), [])) * min(1.0, 1.0 / wall)))
        self.samples = {k: v[-keep:] for k, v in self.samples.items()}
        out = {"card": self.card, "how": "amdgpu hwmon (freq1_input / freq2_input / power1_input / temp2_input) polled every ~15 ms by a thread while "
                      "%d further f_train steps ran for %.1f s after the timed region; min / median / max over the last second (the "
                      "SMU telemetry is a moving average)" % (n, wall), "samples": keep}
        for k, v in self.samples.items():
            if v:
                v = sorted(v)
                out[k] = {"min": round(v[0], 1), "median": round(v[len(v) // 2], 1), "max": round(v[-1], 1)} if k != 'power_cap_w' else round(v[0], 1)
        try:
            out["power_dpm_force_performance_level"] = open(self.level).read().strip()
        except OSError:
            pass
        return out


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def spmm_algorithmic_bytes(n_rows_out, n_cols, nnz, F, b_bytes=4):
    """every operand touched exactly once (fp32 values, int32 indices); the gathered operand B is
    fp32, or bf16 (b_bytes = 2) in the bf16 configuration."""
    return 8 * nnz + 4 * (n_rows_out + 1) + b_bytes * n_cols * F + 4 * n_rows_out * F


def _cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return ''


def cpu_baseline(shape, A, X, Y, tr, dev, hid, C, sample, multithreaded=False):
    """Oracle timed on the host: one f_train step ('step') or one conv layer fwd+bwd ('layer')."""
    from oracle import gcn_oracle as O
    import contextlib
    nnz = A.nnz
    threads = os.cpu_count() or 1
    ctx = contextlib.nullcontext()
    spmm_note = "scipy CSR SpMM is single-threaded like Theano's StructuredDot"
    if multithreaded:
        import scipy.sparse as sps
        from oracle import cpu_mt
        cpu_mt.build()
        At = A if abs(A - A.T).max() == 0 else sps.csr_matrix(A.T)
        ctx = cpu_mt.patched(O, {id(A): At, id(X): sps.csr_matrix(X.T)})           # transposes prepared once, untimed
        spmm_note = "sparse products on %d OpenMP threads (oracle/cpu_mt.c)" % cpu_mt.threads()
    if sample == 'layer':
        rng = np.random.RandomState(1)
        H = rng.randn(A.shape[0], 300).astype(np.float32)
        G = rng.randn(A.shape[0], 300).astype(np.float32)
        W = (rng.randn(300, 300) * 0.05).astype(np.float32)
        b = np.zeros(300, np.float32)
        with ctx:
            t0 = time.time()
            O.conv_layer_fwd_bwd(H, W, b, A, G)
            t = time.time() - t0
        return {"value": nnz / t, "unit": "edges/s", "cores": threads, "cpu_model": _cpu_model(), "kind": "port", "seconds": round(t, 2),
                "sample": "1 ConvolutionDenseLayer2 fwd+bwd (300->300) on the full %s graph; %s, BLAS sgemm uses %d "
                          "threads" % (shape, spmm_note, threads)}
    params = O.random_params(X.shape[1], hid, C, True, seed=7)
    mask = (np.random.RandomState(3).rand(X.shape[0], hid[0]) < 0.5).astype(np.float32)
    st = O.AdamState(params)
    n_conv = len(hid)
    if multithreaded:
        # the "fair multi-core CPU" of SURVEY.md section 8d: EVERY pass of the step on all cores (oracle/cpu_mt.py::f_train:
        # sparse products, fused elementwise / softmax / column-sum passes in OpenMP, dense products in BLAS)
        Xt = sps.csr_matrix(X.T)
        t0 = time.time()
        cpu_mt.f_train(O, params, st, X, Xt, Y[tr], Y[dev], A, At, tr, dev, hid, 0.5, mask)
        t = time.time() - t0
        return {"value": n_conv * nnz / t, "unit": "edges/s", "cores": threads, "cpu_model": _cpu_model(), "kind": "port",
                "seconds": round(t, 2),
                "cores_per_phase": {"sparse products (OpenMP, oracle/cpu_mt.c)": cpu_mt.threads(),
                                    "bias+tanh / highway mix / softmax / gradients / column sums (OpenMP, oracle/cpu_mt.c)": cpu_mt.threads(),
                                    "dense products (BLAS sgemm through NumPy)": threads, "Adam on 3.3 M parameters, loss sums (NumPy)": 1},
                "sample": "1 full f_train step (same workload, same unit: %d conv layers x nnz / step time) on the full %s graph, "
                          "every pass multi-threaded (oracle/cpu_mt.py::f_train; equals oracle.f_train to rounding: "
                          "tests/test_oracle.py); a reported baseline, not the target" % (n_conv, shape)}
    with ctx:
        t0 = time.time()
        O.f_train(params, st, X, Y[tr], Y[dev], A, tr, dev, hid, True, 0.5, mask)
        t = time.time() - t0
    return {"value": n_conv * nnz / t, "unit": "edges/s", "cores": threads, "cpu_model": _cpu_model(), "kind": "port", "seconds": round(t, 2),
            "sample": "1 full f_train step (same workload, same unit: %d conv layers x nnz / step time) on the full %s "
                      "graph; %s, BLAS sgemm uses %d threads" % (n_conv, shape, spmm_note, threads)}


def other_kernels(clf, step, g, hid, N, C, precision, n_steps=5):
    """The other hot kernels timed INSIDE real training steps (ops.StepTimers: an event pair around each wrapper call on the
    launch stream, `n_steps` extra steps after the timed region; median over the calls): the step's own operands, cache state
    and clocks.  HBM-bound ones are priced with their algorithmic bytes (every operand once; in the bf16
    configuration the gathered operand of a graph product is 2 bytes per element), GEMMs with their flops against the dense MFMA
    peak of the precision they run in AND with their operand bytes against HBM (`frac_hbm`): a bf16 product of these shapes is
    nearer to the second roofline than to the first."""
    from geographconv_amd import ops
    F = hid[0]
    bf16 = precision == 'bf16'
    zb = 2 if bf16 else 4                          # bytes per element of the operand a graph product gathers
    with ops.StepTimers() as st:
        for _ in range(n_steps):
            step()
    ms = {k: sorted(v)[len(v) // 2] for k, v in st.ms().items()}
    calls = {k: len(v) // n_steps for k, v in st.ms().items()}
    out = []
    how = "in-step: median of %d calls inside %d f_train steps (event pairs on the launch stream)"

    def hbm(name, key, alg, note=''):
        if key in ms:
            out.append({"kernel": name, "bound": "hbm", "ms": ms[key], "algorithmic_bytes": alg, "achieved": alg / ms[key] / 1e6,
                        "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": alg / ms[key] / 1e6 / HBM_PEAK_GBPS,
                        "calls_per_step": calls[key], "how": how % (calls[key] * n_steps, n_steps), "note": note})

    def mfma(name, key, flops, operand_bytes, prec, note=''):
        if key in ms:
            # split-bf16 products issue SIX bf16 MFMA terms per product: priced with that work against the bf16 peak; the useful
            # (fp32-equivalent) rate stands beside it
            work = 6.0 * flops if prec == 'bf16x3' else flops
            peak = MFMA_F32_PEAK_TFLOPS if prec == 'f32' else MFMA_BF16_PEAK_TFLOPS
            e = {"kernel": name, "bound": "mfma", "precision": prec, "ms": ms[key], "flops": flops, "achieved": work / ms[key] / 1e9,
                 "peak": peak, "unit": "TFLOP/s", "frac": work / ms[key] / 1e9 / peak,
                 "operand_bytes": operand_bytes, "achieved_hbm": operand_bytes / ms[key] / 1e6, "peak_hbm": HBM_PEAK_GBPS,
                 "unit_hbm": "GB/s", "frac_hbm": operand_bytes / ms[key] / 1e6 / HBM_PEAK_GBPS,
                 "calls_per_step": calls[key], "how": how % (calls[key] * n_steps, n_steps), "note": note}
            if prec == 'bf16x3':
                e["mfma_flops_issued"] = work
                e["achieved_fp32_equivalent"] = flops / ms[key] / 1e9
                e["note"] = (note + "; " if note else "") + ("achieved / frac = six bf16 MFMA terms per product against the dense bf16 peak (the socket runs "
                                                             "these kernels at its 1.4 kW power cap, ~1.97 GHz: profiles/r05_x3_rows_clocks.txt); "
                                                             "achieved_fp32_equivalent = useful flops / time")
            out.append(e)
    X, A = g['X'], g['A']
    nnzX, V, nnz = X.fwd.nnz, X.shape[1], A.fwd.nnz
    xw = 8 * nnzX + 4 * (N + 1) + 4 * V * F + 4 * N * F
    hbm("X . W0 + b0, tanh  (spmm_hot_kernel: hot rows of W0 in LDS)", 'spmm_x', xw)
    hbm("X . W0 + b0, tanh, with the dropout that follows in the epilogue  (spmm_hot_kernel<.., DROP>)", 'spmm_x_dropout',
        xw + 4 * N * F + N * F, "algorithmic bytes = the plain product's + the dropped copy and the byte mask written")
    hbm("X^T . dS0  (head panel GEMM + xt_tail_kernel + combine)", 'spmm_t', 8 * nnzX + 4 * N * F + 4 * V * F)
    hbm("tanh(A_hat . Z + bh) with the highway mix in the epilogue  (spmm_rows_kernel<.., HW> + long-row combine)", 'spmm_highway',
        spmm_algorithmic_bytes(N, N, nnz, F, zb) + 3 * 4 * N * F,
        "algorithmic bytes = the plain product's (gathered operand: %d bytes per element) + T, H read and Hout written" % zb)
    fl = 2.0 * N * F * F
    wb = 4 * F * F                                 # one weight matrix
    act = 4 * N * F                                # one N x F fp32 activation
    from geographconv_amd import tuning as _tuning
    fp = precision if precision in ('bf16x3', 'bf16') else 'f32'        # precision of the fused highway launches (bf16: only the k-concatenated dH is one)
    rows = {'bf16x3': "x3_rows_kernel: 64 rows of A per block in K chunks, three bf16 planes in LDS",
            'bf16': "gemm_bf16_rows_kernel: 64 whole rows of both A operands per block as bf16 in LDS",
            'f32': "gemm_rows_kernel: 64 whole rows of A per block"}[fp]
    mfma("H . [Wh | Wt], sigmoid on the gate half  (%s, dual)" % rows, 'gemm_dual_nn', 2 * fl, 3 * act + 2 * wb, fp)
    mfma("H^T . [dZ | dU]  (%s; split-K + ordered combine)" % (
        "x3_tn_kernel: operands transposed + split into three bf16 planes on their way into LDS" if fp == 'bf16x3'
        else "gemm_tn_direct_kernel: fragments straight into registers, no LDS"), 'gemm_dual_tn', 2 * fl, 3 * act + 2 * wb, fp)
    if _tuning.FUSE_GATE_CARRY:
        mfma("dH = dZ . Wh^T + dU . Wt^T + G (1 - T)  (%s, two A operands into one accumulator, the block's carry "
             "gradient formed in the epilogue)" % rows.split(':')[0], 'gemm_kcat', 2 * fl, 5 * act + 2 * wb, fp,
             "operand bytes: dZ, dU, G, T read, dH written (highway_bwd does not store the carry); the median is over both blocks' calls"
             + ("; the first block's also applies the dropout mask and the tanh gradient of the sparse-input layer (reads H0 and the mask)"
                if _tuning.FUSE_ACT_BWD else ""))
    else:
        mfma("dH = dZ . Wh^T + dU . Wt^T [+ carry]  (%s, two A operands into one accumulator)" % rows.split(':')[0], 'gemm_kcat', 2 * fl,
             4 * act + 2 * wb, fp, "operand bytes: dZ, dU read, the carry read and dH written")
    # single products (every GEMM of the bf16 / bf16x3 configurations, the output layer's in all): label = shape and form
    for key in sorted(k for k in ms if k.startswith('gemm:')):
        _, form, prec, M_, N_, K_, cb, acc = key.split(':')
        M_, N_, K_, cb, acc = int(M_), int(N_), int(K_), int(cb), int(acc)
        if 2.0 * M_ * N_ * K_ < 1e9:
            continue
        opb = 4 * (M_ * K_ + K_ * N_) + cb * M_ * N_ * (2 if acc else 1)
        kern = {'f32': 'gemm_tn_direct_kernel' if form == 'tn' else
                       ('gemm_rows_kernel' if (form == 'nt' or N_ in range(193, 257) or N_ in range(449, 513)) and M_ >= 32768 and N_ <= 640
                        and -(-K_ // 16) * 16 in (256, 304) else 'gemm_kernel'),
                'bf16x3': ('x3_tn_kernel' if form == 'tn' and N_ > 160 else 'x3_rows_kernel' if form != 'tn' and M_ >= 4096 and N_ <= 1024 and K_ <= 1024
                           else 'exact fp32 kernel'),
                'bf16': 'gemm_bf16_rows_kernel / gemm_bf16_kernel / gemm_bf16_tn_kernel'}[prec]
        mfma("%s product %d x %d x %d (%s; C %s%s)" % ({'nn': 'A . B', 'nt': 'A . B^T', 'tn': 'A^T . B'}[form], M_, N_, K_, kern,
                                                         'bf16' if cb == 2 else 'fp32', ', accumulating' if acc else ''),
             key, 2.0 * M_ * N_ * K_, opb, prec if kern != 'exact fp32 kernel' else 'f32')
    return out


def baseline_config(shape, hid, precision, world):
    """Which entry of BASELINE.json `configs` this invocation is -- or that it is none of them."""
    hid = list(hid)
    # (configs[1..3] name an fp32 model: 'f32' = exact fp32 products, 'bf16x3' = fp32-class split-bf16 products -- config.gemm says which)
    if shape == 'cmu' and hid == [300, 300, 300] and precision in ('f32', 'bf16x3') and world == 1:
        return 'BASELINE configs[1]'
    if shape == 'twus' and hid == [300, 300, 300] and precision in ('f32', 'bf16x3'):
        return 'BASELINE configs[2]' if world == 1 else 'BASELINE configs[3]'
    if shape == 'twus' and hid == [600] * 6 and precision == 'bf16':
        return 'BASELINE configs[4]' + ('' if world == 8 else ' on %d GPU%s instead of 8' % (world, '' if world == 1 else 's'))
    return 'NOT a BASELINE config: a variant (shape %s, hid %s, GEMM precision %s)' % (shape, 'x'.join(map(str, hid)), precision)


def _free_port():
    """A port for the rendezvous: below the range the kernel hands to outgoing connections (32768-60999 by default), so that none of the
    job's own connects -- gloo / RCCL bootstrap pairs use ephemeral ports -- can take it between this probe and the launcher's bind."""
    import random
    import socket
    rng = random.Random(os.getpid() ^ int(time.time() * 1e3))
    for _ in range(64):
        port = rng.randrange(20000, 32000)
        with socket.socket() as so:
            try:
                so.bind(('127.0.0.1', port))
                return port
            except OSError:
                continue
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0))
        return so.getsockname()[1]


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU, the same
    torch.distributed.run command the driver contract names) and hand their output through; rank 0 prints the line."""
    import subprocess
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    rc = 1
    for attempt in (0, 1):
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
               '--master-port', str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        log('[bench] --gpus %d without WORLD_SIZE: launching %s' % (n, ' '.join(cmd[1:9])))
        t0 = time.time()
        rc = subprocess.run(cmd, env=env).returncode
        # a launch that dies within seconds never got past the rendezvous (the port found free a moment ago was taken, a listener
        # refused): one more try on another port; anything later is the job's own result
        if rc == 0 or attempt == 1 or time.time() - t0 > 30.0:
            break
        log('[bench] the launch ended with code %d after %.0f s: once more on another port' % (rc, time.time() - t0))
    raise SystemExit(rc)


def timed_region(clf, step, steps, warmup, barrier, F_spmm, g):
    """W untimed steps, then exactly K steps between barrier + synchronize pairs; the SpMM timer (hipEvent pairs recorded
    by the library on the launch stream) samples the plain F_spmm-wide graph products inside the region."""
    import torch
    from geographconv_amd import ops
    for _ in range(warmup):
        step()
    timer = ops.SpmmTimer(capacity=max(16, 8 * steps))
    timer.attach(g['A'].fwd, only_F=F_spmm)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    # (a generation-2 collection of the interpreter takes 10-20 ms with the graph's host arrays alive: it would stall the
    #  launching thread in the middle of a step -- one 49 ms step among 28 ms ones was seen once; collect now, not in the region)
    import gc
    gc.collect()
    gc_was_on = gc.isenabled()
    gc.disable()
    barrier()
    t0 = time.perf_counter()
    last = None
    marks[0].record()
    for i in range(steps):
        last = step()
        marks[i + 1].record()
    barrier()
    t = time.perf_counter() - t0
    if gc_was_on:
        gc.enable()
    timer.detach()
    kern_ms = timer.read_ms()
    step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(steps))
    return t, step_ms, kern_ms, last


def partition_check(make_clf, comms, X, A, Y, tr, dev, rank, n_steps=2):
    """Parity evidence carried by an N > 1 line: `n_steps` training steps with dropout 0 through every exchange scheme,
    compared on rank 0 with the SAME process's un-partitioned steps (same seed => same initial parameters): losses, hit
    counts, the gathered probabilities, argmax agreement."""
    import torch.distributed as dist

    def run(comm):
        clf = make_clf(comm, 0.0)
        out = []
        for _ in range(n_steps):
            o = clf.f_train(X, Y[tr], Y[dev], A, tr, dev)
            out.append(([float(v) for v in o[:4]], clf.gather_output(o[4])))
        return out
    got = {name: run(c) for name, c in comms.items()}
    dist.barrier()
    res = None
    if rank == 0:
        want = run(None)
        res = {"how": "%d f_train steps, dropout 0, every scheme vs this process's un-partitioned steps on rank 0 (same seed)" % n_steps}
        for name, g in got.items():
            dl = max(abs(a - b) for (gs, _), (ws, _) in zip(g, want) for a, b in ((gs[0], ws[0]), (gs[2], ws[2])))
            dacc = max(abs(a - b) for (gs, _), (ws, _) in zip(g, want) for a, b in ((gs[1], ws[1]), (gs[3], ws[3])))
            dP = max(float(np.abs(gp - wp).max()) for (_, gp), (_, wp) in zip(g, want))
            agree = min(float((gp.argmax(1) == wp.argmax(1)).mean()) for (_, gp), (_, wp) in zip(g, want))
            res[name] = {"max_abs_dloss": dl, "max_abs_dacc": dacc, "max_abs_dP": dP, "argmax_agreement": agree,
                         "losses_partitioned": [gs[0] for gs, _ in g], "losses_single": [ws[0] for ws, _ in want]}
    dist.barrier()
    return res


def layer_block(g, N, nnz, F, device, precision, reps=50):
    """SURVEY.md section 8d (ii) on the GPU: ONE ConvolutionDenseLayer2 forward + backward (300 -> 300; gcnmodel.py:114-136 and what
    autodiff derives from it) and ONE highway block (gcnmodel.py:268-288) on the full graph, the launches the model makes for them,
    event-timed over `reps` repetitions after 3 warm-up ones; edges/s = nnz(A_hat) / time of one forward + backward."""
    import torch
    from geographconv_amd import ops
    A = g['A']
    rng = np.random.RandomState(11)

    def dm(scale=1.0):
        return ops.DMat.from_numpy((rng.randn(N, F) * scale).astype(np.float32), device)
    H, G = dm(), dm(1e-3)
    W = ops.DMat.from_numpy((rng.randn(F, F) * 0.05).astype(np.float32), device)
    W2 = ops.DMat.from_numpy((rng.randn(F, F) * 0.05).astype(np.float32), device)
    b = torch.zeros(ops.pad4(F), device=device)
    db, db2 = torch.zeros_like(b), torch.zeros_like(b)
    dW, dW2 = W.like(), W2.like()
    Z = ops.DMat.empty(N, F, device, ld=ops.gather_ld(F))

    def conv_layer():
        ops.gemm(H, W, out=Z, precision=precision)                                  # Z = H . W
        S = ops.spmm(A.fwd, Z, bias=b, act=ops.ACT_TANH)                            # S = tanh(A_hat . Z + b)
        dS = ops.act_bwd_colsum(G, S, ops.ACT_TANH, db, out=ops.DMat.empty(N, F, device, ld=ops.gather_ld(F)))
        dZ = ops.spmm(A.bwd, dS)                                                    # A_hat^T . dS
        ops.gemm(H, dZ, out=dW, transA=True, precision=precision)                   # dW = H^T . dZ
        return ops.gemm(dZ, W, transB=True, precision=precision)                    # dH = dZ . W^T

    def highway_block():
        _, T = ops.gemm_dual(H, W, W2, out0=Z, bias1=b, act1=ops.ACT_SIGMOID, precision=precision)      # Z = H . Wh, T = sigmoid(H . Wt + bt)
        Hc, Hout = ops.spmm_highway(A.fwd, Z, b, T, H)                              # Hc = tanh(A_hat . Z + bh), Hout = T Hc + (1 - T) H
        dS, dU, _ = ops.highway_bwd(G, T, Hc, H, dbS=db, dbU=db2, carry=False)
        dZ = ops.spmm(A.bwd, dS)
        ops.gemm_dual(H, dZ, dU, out0=dW, out1=dW2, transA=True, precision=precision)
        return ops.gemm_kcat(dZ, W, dU, W2, transB=True, gate_carry=ops.GateCarry(G, T), precision=precision)

    out = {"how": "the launches GraphConv makes for the layer, forward then backward, event-timed over %d repetitions after 3 warm-up "
                  "ones on the full graph (N = %d, %d -> %d, GEMM precision %s); edges/s = nnz(A_hat) / time" % (reps, N, F, F, precision)}
    for name, fn in (("conv_layer_fwd_bwd", conv_layer), ("highway_block_fwd_bwd", highway_block)):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / reps
        out[name] = {"ms": ms, "value": nnz / ms * 1e3, "unit": "edges/s"}
    out["conv_layer_fwd_bwd"]["what"] = ("ConvolutionDenseLayer2 (gcnmodel.py:114-136): Z = H.W, S = tanh(A_hat.Z + b); backward: dS = G (1 - S^2) "
                                         "+ bias gradient, dZ = A_hat^T.dS, dW = H^T.dZ, dH = dZ.W^T")
    out["highway_block_fwd_bwd"]["what"] = ("highway_dense (gcnmodel.py:268-288): (Z, T) = (H.Wh, sigmoid(H.Wt + bt)), Hout = T tanh(A_hat.Z + bh) + (1 - T) H; "
                                            "backward: the gating layer's three gradients + both bias gradients, dZ = A_hat^T.dS, (dWh, dWt) = H^T.[dZ | dU], "
                                            "dH = dZ.Wh^T + dU.Wt^T + G (1 - T)")
    return out


def gather_ceiling(table_mb=512, row_bytes=1280):
    """The rate at which this GPU can gather `row_bytes`-byte rows from a table beyond L2 (tools/micro/gather_bw.hip, built by
    __graft_entry__.build() into tools/micro/bin/): what a graph product on the pinned graph -- half of whose column references
    are uniformly random -- can move at most.  Measured live; None when the helper is not built."""
    import subprocess
    exe = os.path.join(ROOT, 'tools', 'micro', 'bin', 'gather_bw')
    if not os.path.exists(exe):
        return None
    try:
        r = subprocess.run([exe, '--json', str(table_mb), str(row_bytes)], capture_output=True, text=True, timeout=120)
        d = json.loads(r.stdout.strip().splitlines()[-1])
        d["source"] = "tools/micro/gather_bw.hip run by this invocation (16-lane groups, %d-byte rows drawn uniformly from a %d MB table)" % (row_bytes, table_mb)
        return d
    except Exception as e:
        return {"error": repr(e)}


def pmc_per_dispatch(directory, counter, kernel_substr):
    """{(file, dispatch id): counter value} over every rocprofv3 counter table under `directory`, for the dispatches of kernels whose name
    contains `kernel_substr` (a dispatch's rows -- one per counter instance when the tool splits them -- are summed)."""
    import csv
    import glob
    per_dispatch = {}
    for f in glob.glob(os.path.join(directory, '**', '*counter_collection.csv'), recursive=True):
        for row in csv.DictReader(open(f)):
            if row['Counter_Name'] == counter and kernel_substr in row['Kernel_Name']:
                key = (f, row['Dispatch_Id'])
                per_dispatch[key] = per_dispatch.get(key, 0.0) + float(row['Counter_Value'])
    return per_dispatch


def traffic_live(child_args, kernel_substr, timeout=300.0):
    """HBM-side bytes per launch of the dominant kernel, counted by THIS invocation: two `rocprofv3 --pmc` child passes (FETCH_SIZE, then
    WRITE_SIZE -- one counter per pass, nothing else traced, as /opt/skills/guides/MI355X_MICROARCH.md prescribes) of this script with
    `child_args` (the same job, 2 timed steps).  gfx950 correction of that guide: FETCH_SIZE tallies 128-byte read requests as 64 bytes,
    so the read side is 2 x FETCH_SIZE.  Returns a dict with `bytes` or with `error` (the caller then falls back to the stored pass)."""
    import shutil
    import subprocess
    import tempfile
    if 'rocprofiler' in os.environ.get('LD_PRELOAD', '') or any(k.startswith(('ROCPROF', 'ROCP_')) for k in os.environ):
        return {"error": "this invocation runs under a profiler itself: no nested counter passes"}
    exe = shutil.which('rocprofv3') or '/opt/rocm/bin/rocprofv3'
    if not os.path.exists(exe):
        return {"error": "rocprofv3 not found"}
    tmp = tempfile.mkdtemp(prefix='geogcn_pmc_', dir='/tmp')
    kb, n_launch, secs = {}, {}, {}
    try:
        for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
            d = os.path.join(tmp, counter)
            cmd = [exe, '--pmc', counter, '--output-format', 'csv', '-d', d, '-o', 'p', '--', sys.executable, os.path.abspath(__file__)] + child_args
            t0 = time.time()
            r = subprocess.run(cmd, cwd='/tmp', env=dict(os.environ, TMPDIR='/tmp'), capture_output=True, text=True, timeout=timeout)
            secs[counter] = round(time.time() - t0, 1)
            per_dispatch = pmc_per_dispatch(d, counter, kernel_substr)
            if not per_dispatch:
                return {"error": "pass %s: no dispatch of %s in the counter table (child rc %d: %s)" % (counter, kernel_substr, r.returncode, (r.stderr or '')[-300:])}
            kb[counter] = sum(per_dispatch.values()) / len(per_dispatch)
            n_launch[counter] = len(per_dispatch)
        # (the factor 2 is the guide's calibration for wide coalesced reads -- 16 B per lane, which is how this kernel reads rows of B,
        #  indices and values; narrower requests would be over-counted by it, so the uncorrected sum travels with the corrected one)
        return {"bytes": (2.0 * kb['FETCH_SIZE'] + kb['WRITE_SIZE']) * 1024.0, "fetch_size_kb": kb['FETCH_SIZE'], "write_size_kb": kb['WRITE_SIZE'],
                "bytes_uncorrected": (kb['FETCH_SIZE'] + kb['WRITE_SIZE']) * 1024.0, "read_factor": 2.0,
                "read_factor_is": "an assumption: every read request 128 B tallied as 64 B (MI355X_MICROARCH.md, calibrated on 16-B-per-lane streams)",
                "launches_counted": n_launch, "pass_seconds": secs}
    except Exception as e:
        return {"error": repr(e)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def _git_commit():
    import subprocess
    try:
        return subprocess.run(['git', '-C', ROOT, 'rev-parse', '--short', 'HEAD'], capture_output=True, text=True, timeout=10).stdout.strip() or None
    except Exception:
        return None


class Watchdog:
    """Per-scheme wall-clock guard of an N > 1 invocation: a collective that never returns cannot be interrupted from Python, so when
    the timer fires rank 0 prints the line it has (the schemes already timed, the guarded one reported as an error) and EVERY rank
    leaves with os._exit -- the headline survives a hang in a scheme timed after it.  Exit code 0 on purpose (ADVICE round 5 asked for a
    non-zero code from the other ranks): a launcher that sees a failed rank tears the job down and a driver that sees a failed job drops
    its line; the timeout is reported IN the line instead -- the stuck scheme's entry carries `"error": "timeout: ..."` and the line
    `"watchdog_fired": true` (tests/test_dist_gpu.py::test_bench_survives_a_scheme_that_hangs)."""

    def __init__(self, rank):
        self.rank, self.timer, self.on_fire = rank, None, None

    def arm(self, seconds, what, on_fire):
        import threading
        self.disarm()

        def fire():
            log('[bench] rank %d: %s did not finish within %.0f s: leaving with what has been measured' % (self.rank, what, seconds))
            try:
                if self.rank == 0:
                    on_fire("timeout: not finished within %.0f s (wall-clock guard)" % seconds)
            finally:
                sys.stdout.flush()
                sys.stderr.flush()
                os._exit(0)
        self.timer = threading.Timer(seconds, fire)
        self.timer.daemon = True
        self.timer.start()

    def disarm(self):
        if self.timer is not None:
            self.timer.cancel()
            self.timer = None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--shape', default='twus', choices=['twus', 'cmu', 'twus_sbm'])
    ap.add_argument('--reorder', default=None, choices=['degree', 'rcm', 'bfs', 'lpa', 'auto'],
                    help='node renumbering applied on the device side (outputs stay in original ids)')
    ap.add_argument('--hid', nargs='+', type=int, default=[300, 300, 300])
    ap.add_argument('--dropout', type=float, default=0.5)
    ap.add_argument('--gemm-precision', default=None, choices=['f32', 'bf16x3', 'bf16'],
                    help="default: geographconv_amd.tuning.GEMM_PRECISION (bf16x3 = fp32-class split-bf16 products where a kernel takes the "
                         "shape, exact fp32 elsewhere); f32 = the exact fp32 MFMA everywhere")
    ap.add_argument('--cpu-sample', default='step', choices=['step', 'layer', 'none'])
    ap.add_argument('--exchange', default='auto', choices=['auto', 'a2a', 'allgather', 'agpipe', 'halo'],
                    help="N > 1: the exchange scheme timed FIRST (auto: the north_star's all-gather); the others are timed after it, each "
                         "behind its own try/except and wall-clock guard, and `value` is the fastest that finished")
    ap.add_argument('--no-alt', action='store_true', help='N > 1: time only the first scheme')
    ap.add_argument('--schemes', default='allgather,a2a,agpipe',
                    help='N > 1: the exchange schemes timed (and checked) after the first one, comma-separated (default: all of them)')
    ap.add_argument('--no-check', action='store_true', help='N > 1: skip the partition_check block')
    ap.add_argument('--graph', action='store_true',
                    help='N > 1: time the fastest CAPTURABLE scheme (the all-gather ones) once more with its step captured in a hipGraph; off by '
                         'default: one simulated rank gains 1 %% (profiles/r06_sim_rank_bf16x3.txt) and RCCL capture has only ever run at world 1 here')
    ap.add_argument('--no-extras', action='store_true', help='N = 1: skip alt_exact_f32, the layer block and the gather ceiling')
    ap.add_argument('--scheme-timeout', type=float, default=None,
                    help='N > 1: wall-clock guard per scheme / check, seconds (default 300; 900 under the host-staged functional-check backend)')
    ap.add_argument('--traffic', default=None, choices=['live', 'stored', 'none'],
                    help="roofline.traffic of the headline configuration at N = 1: live = two rocprofv3 --pmc child passes of this command run by "
                         "this invocation (+ ~1 min; falls back to `stored` when they cannot run; the default without --no-extras); "
                         "stored = the committed pass under profiles/ (the default with --no-extras)")
    ap.add_argument('--pmc-child', action='store_true', help=argparse.SUPPRESS)      # the child of a `--traffic live` pass: the timed steps only
    ap.add_argument('--set', action='append', default=[], metavar='NAME=VALUE',
                    help='A/B aid: override an attribute of geographconv_amd/tuning.py for this run (e.g. --set FUSE_CARRY=0); '
                         'recorded in config.tuning_overrides -- a line with overrides is not the headline configuration')
    args = ap.parse_args()
    if args.pmc_child:
        args.no_extras, args.cpu_sample, args.traffic = True, 'none', 'none'
    if args.traffic is None:
        args.traffic = 'stored' if args.no_extras else 'live'

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        self_launch(args.gpus)

    import torch
    from geographconv_amd import ops, synth, tuning
    from geographconv_amd.gcnmodel import GraphConv
    overrides = {}
    for kv in args.set:
        name, _, val = kv.partition('=')
        if not hasattr(tuning, name):
            raise SystemExit("--set: geographconv_amd.tuning has no attribute %r" % name)
        old = getattr(tuning, name)
        new = (val not in ('0', 'false', 'False', '')) if isinstance(old, bool) else type(old)(val)
        setattr(tuning, name, new)
        overrides[name] = new
    precision = args.gemm_precision or ops.GEMM_PRECISION

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
    ops.require_gpu()
    # GEOGCN_DIST_BACKEND=staged-gloo: functional check of the N > 1 branches on a ONE-GPU box (never a measurement): all
    # ranks share cuda:0 and the collectives are staged through the host and gloo (geographconv_amd/dist.py)
    from geographconv_amd import dist as gdist
    staged = gdist.backend_name() == 'staged-gloo'
    if args.scheme_timeout is None:
        # a healthy scheme is a model build + W + K steps of tens of milliseconds: well under a minute on real links
        args.scheme_timeout = 900.0 if staged else 300.0
    force_dist = os.environ.get('GEOGCN_BENCH_FORCE_DIST') == '1'      # exercise the partitioned path at world 1
    distributed = world > 1 or force_dist
    if distributed:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        device = gdist.init_process_group(local)
    else:
        torch.cuda.set_device(local)
        device = torch.device('cuda', local)
    # test hook: GEOGCN_BENCH_INJECT="a2a:raise;agpipe:hang" makes the named scheme fail that way on every rank (tests/test_dist_gpu.py)
    inject = dict(kv.split(':') for kv in os.environ.get('GEOGCN_BENCH_INJECT', '').split(';') if ':' in kv)

    # who is here: world size as the process group sees it, the collective library's version, every rank's device
    dist_info = None
    if distributed:
        import torch.distributed as tdist
        dist_info = {"world_size_seen": tdist.get_world_size(), "torch_backend": tdist.get_backend(), "staged": staged}
        try:
            pr = torch.cuda.get_device_properties(device)
            mine = {"rank": rank, "local_rank": local, "device": str(device), "name": pr.name,
                    "pci": "%04x:%02x:%02x" % (getattr(pr, 'pci_domain_id', 0), getattr(pr, 'pci_bus_id', 0), getattr(pr, 'pci_device_id', 0))}
            everyone = [None] * tdist.get_world_size()
            tdist.all_gather_object(everyone, mine)
            dist_info["ranks"] = everyone
        except Exception as e:
            dist_info["ranks_error"] = repr(e)
        try:
            dist_info["rccl_version"] = '.'.join(map(str, torch.cuda.nccl.version()))
        except Exception as e:
            dist_info["rccl_version"] = repr(e)

    t0 = time.time()
    A, X, Y, (tr, dev, te), C = synth.make_graph(args.shape)
    N, nnz = A.shape[0], int(A.nnz)
    if rank == 0:
        log('[bench] %s graph generated in %.1fs: N=%d nnz(A)=%d nnz(X)=%d' % (args.shape, time.time() - t0, N, nnz, X.nnz))

    def make_comm(exchange):
        from geographconv_amd.dist import TorchDistComm
        return TorchDistComm(N, device, exchange=exchange)

    def make_clf(c, dropout, prec=None, hip_graph=False):
        m = GraphConv(X.shape[1], C, args.hid, 0.0, dropout, highway=True, device=device, comm=c,
                      gemm_precision=prec or precision, reorder=args.reorder, hip_graph=hip_graph)
        m.build_model(A, seed=77)
        m._force_dist = force_dist and c is not None
        return m

    y_tr, y_dev = Y[tr], Y[dev]
    tr, dev = np.array(tr), np.array(dev)
    for v in (y_tr, y_dev, tr, dev):
        # the caller's promise that these never change: GraphConv then hashes each once instead of every step
        # (gcnmodel._content_key; ~0.1 ms per MB and step otherwise)
        v.setflags(write=False)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def reduce_max(t):
        if world > 1:
            tt = torch.tensor([t], dtype=torch.float64, device='cpu' if staged else device)
            torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
            t = float(tt.item())
        return t

    bf16_operand = precision == 'bf16'
    n_conv = len(args.hid)
    live_traffic = {}          # filled by the `--traffic live` passes (N = 1, headline configuration) before the line is built

    def run_scheme(c, prec=None, hip_graph=False):
        """W warm-up + K timed steps of one configuration -> everything the line needs from it.  `hip_graph`: the step captured in a
        hipGraph (two eager steps, the capture, then replays: at least four untimed steps before the timed region)."""
        how = inject.get((c.exchange if c is not None else 'single') + ('+hipgraph' if hip_graph else ''))
        if how == 'raise':
            raise RuntimeError("injected failure (GEOGCN_BENCH_INJECT) in scheme %s" % c.exchange)
        if how == 'hang':
            while True:
                time.sleep(1.0)
        m = make_clf(c, args.dropout, prec, hip_graph)
        # the dense operand of the timed SpMM: F = hid at one GPU; one feature panel per rank under the a2a scheme
        F = args.hid[-1]
        if c is not None and c.exchange == 'a2a':
            F = c.panel_width(args.hid[-1], bf16_operand)
        g = m._device_graph(X, A)
        t, step_ms, kern_ms, last = timed_region(m, lambda: m.f_train(X, y_tr, y_dev, A, tr, dev), args.steps,
                                                 max(args.warmup, 4) if hip_graph else args.warmup, barrier, F, g)
        if hip_graph and not (m._hg is not None and m._hg.get('graph') is not None):
            raise RuntimeError("the step was not captured (transport not capturable, or the capture was dropped)")
        return {"comm": c, "clf": m, "g": g, "F": F, "t": reduce_max(t), "step_ms": step_ms, "kern_ms": kern_ms, "last": last,
                "hip_graph": bool(hip_graph)}

    notes = {'allgather': "the north_star's 1-D row split of A_hat + all-gather of H",
             'agpipe': "the north_star's 1-D row split of A_hat + all-gather of H in feature slabs, slab q + 1 on the wire while slab q is multiplied",
             'a2a': "feature repartition with two all-to-alls", 'halo': "halo exchange"}

    def summary(name, r):
        return {"exchange": name, "value": n_conv * nnz * args.steps / r["t"], "unit": "edges/s", "ms_per_step": r["t"] / args.steps * 1e3,
                "step_ms_median": r["step_ms"][len(r["step_ms"]) // 2], "train_loss_last": float(r["last"][0]),
                "note": "same job, same K/W, exchange scheme: %s" % notes.get(name, name)}

    def build_line(r, extras):
        """The JSON line with scheme result `r` as the headline (rank 0)."""
        comm, g, F = r["comm"], r["g"], r["F"]
        t, step_ms, kern_ms, last = r["t"], r["step_ms"], r["kern_ms"], r["last"]
        csr = g['A'].fwd
        deg = np.diff(csr.rowptr_host)
        # one product = spmm_rows_kernel over every stored edge of the local row block (short rows with the fused epilogue
        # + the 128-nonzero chunks of the long rows) + the long rows' ordered combine: algorithmic bytes = SURVEY.md §8d
        alg = spmm_algorithmic_bytes(len(deg), csr.shape[1], int(csr.nnz), F, 2 if bf16_operand else 4)
        avg_ms = float(np.mean(kern_ms)) if kern_ms else None
        achieved = alg / (avg_ms * 1e-3) / 1e9 if kern_ms else None
        traffic, traffic_source = None, None
        pmc_file = os.path.join(ROOT, 'profiles', 'pmc_spmm_bf16_latest.json' if bf16_operand else 'pmc_spmm_latest.json')
        if live_traffic.get("bytes"):
            traffic = live_traffic["bytes"]
            traffic_source = ("measured by this invocation: two rocprofv3 --pmc child passes (FETCH_SIZE, then WRITE_SIZE; one counter per pass) of this "
                              "command with --steps 2 --warmup 1, averaged over the %s launches of %s they saw; 2 x FETCH_SIZE (gfx950: 128-byte "
                              "requests tallied as 64) + WRITE_SIZE per launch; the counters tally L2 -> fabric requests, Infinity-Cache hits included"
                              % (live_traffic["launches_counted"], live_traffic["kernel"]))
        elif args.traffic != 'none' and os.path.exists(pmc_file) and world == 1 and F == 300 and args.shape == 'twus' and args.reorder is None:
            try:
                pm = json.load(open(pmc_file))
                traffic = pm.get('hbm_bytes_per_launch')
                traffic_source = ("NOT measured in this run: rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE separately) of this command "
                                  "at state %s (commit %s; table %s), %s; 2 x FETCH_SIZE + WRITE_SIZE per launch; the counter tallies "
                                  "L2 -> fabric requests, Infinity-Cache hits included" % (pm.get('state'), pm.get('commit'), pm.get('summary'),
                                                                                         os.path.relpath(pmc_file, ROOT)))
                if live_traffic.get("error"):
                    traffic_source += "  [the live passes of this invocation did not run: %s]" % live_traffic["error"]
            except Exception:
                traffic = None
        roofline = {"bound": "hbm", "kernel": "spmm_rows_kernel<%d,*> + spmm_long_reduce_kernel (A_hat^T . dS, F=%d%s)" % (
                        (F + 63) // 64, F, "" if comm is None else (", rank 0's feature panel of all rows" if comm.exchange == 'a2a'
                                                                    else ", rank 0's row block against the gathered operand")),
                    "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": (achieved / HBM_PEAK_GBPS) if achieved else None, "traffic": traffic, "traffic_source": traffic_source,
                    "algorithmic_bytes_per_launch": alg, "avg_launch_ms": avg_ms, "launches_timed": len(kern_ms),
                    "launches_sampled": "every plain (not highway-fused) graph product with F = %d over all %d stored edges inside "
                                        "the timed region (the backward products of the hidden layers), hipEvent pairs recorded by "
                                        "the library on the launch stream around row kernel + long-row combine" % (F, int(csr.nnz)),
                    "edges_per_launch": int(csr.nnz), "bytes_per_edge": alg / max(1, int(csr.nnz))}
        if live_traffic.get("bytes"):
            roofline["traffic_counters"] = {k: live_traffic[k] for k in ("fetch_size_kb", "write_size_kb", "bytes_uncorrected", "read_factor", "read_factor_is", "launches_counted", "pass_seconds") if k in live_traffic}
        roofline.update(extras.get("roofline", {}))
        nnz_bwd_out = int(g['A_tr'][1].nnz) if g.get('A_tr') is not None else nnz
        di = None
        if dist_info is not None:
            di = dict(dist_info, exchange=comm.exchange + ('+hipgraph' if r.get("hip_graph") else ''),
                      data_path=type(comm.dist).__name__ if hasattr(comm, 'dist') and not isinstance(comm.dist, type(torch.distributed)) else "torch.distributed")
        out = {
            "metric": "GCN-layer fwd+bwd edges/sec", "value": n_conv * nnz * args.steps / t, "unit": "edges/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": t / args.steps * 1e3,
            "step_ms": {"median": step_ms[len(step_ms) // 2], "p10": step_ms[int(0.1 * (len(step_ms) - 1))],
                        "p90": step_ms[int(round(0.9 * (len(step_ms) - 1)))], "min": step_ms[0], "max": step_ms[-1],
                        "how": "torch.cuda.Event after every step of the timed region (device time between consecutive steps)"},
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32" if precision != "bf16" else "bf16", "data": "synthetic",
            "config": {"workload": "%s synthetic CSR (%s): N=%d, nnz(A_hat)=%d, "
                                   "X %dx%d nnz=%d, C=%d; %s highway GCN, dropout %.2f, Adam; full-graph f_train step"
                                   % ({'twus': 'TwitterUS-shape power-law', 'cmu': 'CMU-shape power-law',
                                       'twus_sbm': 'TwitterUS-size community-structured'}[args.shape],
                                      baseline_config(args.shape, args.hid, precision, world), N, nnz, N, X.shape[1], X.nnz, C,
                                      'x'.join(map(str, args.hid)), args.dropout),
                       "edges_per_step": n_conv * nnz,
                       "edges_traversed_per_step": 2 * (n_conv - 1) * nnz + nnz + nnz_bwd_out,
                       "parallelism": "rows%d" % world if world > 1 else "single",
                       "hip_graph": bool(r.get("hip_graph")),
                       "world_size": world, "collectives": None if comm is None else ("%s, exchange = %s" % ("STAGED through the host + gloo (functional check, NOT a measurement)" if staged else "RCCL (torch.distributed nccl)", comm.exchange)),
                       "dist": di,
                       "reorder": args.reorder,
                       "tuning_overrides": overrides or None,
                       "dropout_stream": "Philox keyed by device row: with --reorder the dropped entries differ from the "
                                         "un-reordered run of the same seed (statistically equivalent, not bitwise)" if args.reorder else "Philox",
                       "gemm": {"f32": "exact fp32 MFMA (v_mfma_f32_16x16x4_f32) in every product",
                                "bf16x3": "bf16x3 split MFMA, fp32 accumulate (fp32-class): every fp32 operand split exactly into three bf16 terms, six "
                                          "v_mfma_f32_16x16x32_bf16 cross terms per product (csrc/gemm_x3.hip) where a kernel takes the shape -- "
                                          "the TwitterUS-size products -- exact fp32 MFMA elsewhere; inputs, outputs and accumulation fp32; the GPU "
                                          "suite holds it to the tolerances stated for the exact kernels (0 argmax mismatches over 440,000 rows)",
                                "bf16": "bf16 MFMA, fp32 accumulate"}[precision],
                       "train_loss_last": float(last[0])},
            "roofline": roofline,
        }
        for k in ("clocks", "alt_exact_f32", "layer", "exchange_choice", "alt", "alt2", "partition_check", "cpu_baseline", "cpu_baseline_mt"):
            if k in extras:
                out[k] = extras[k]
        return out

    def emit(out):
        # RCCL prints a version banner through C stdio; flush it so that the JSON line is the LAST line of stdout
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(out), flush=True)

    # ---- the headline scheme: at N > 1 the north_star's all-gather, timed FIRST; its line is printed before anything else can fail ------
    watchdog = Watchdog(rank)
    first_name = None
    comm = None
    if distributed:
        first_name = 'allgather' if args.exchange == 'auto' else args.exchange
        comm = make_comm(first_name)
    results = {}
    first = run_scheme(comm)
    results[first_name] = first
    extras = {}
    multi = world > 1 or force_dist          # (GEOGCN_BENCH_FORCE_DIST=1: the whole N > 1 flow at world 1 -- real RCCL on a one-GPU box; tests/test_dist_gpu.py)
    if multi and rank == 0:
        early = build_line(first, {})
        early["note"] = ("early line: the first scheme only, printed before the other schemes and the partition check run -- a later line of "
                         "this invocation supersedes it")
        emit(early)

    # ---- N > 1: the other schemes, each behind its own try/except and wall-clock guard ----------------------------------------------------
    failed = []

    def final_line(pending_error=None):
        done = {k: v for k, v in results.items() if k is not None}
        best = min(done, key=lambda e: done[e]["t"]) if done else None
        head = done[best] if best is not None else first
        ex = dict(extras)
        if multi:
            others = [summary(k, v) for k, v in done.items() if k != best] + failed + ([pending_error] if pending_error else [])
            ex["exchange_choice"] = {"how": "the north_star's all-gather timed first; every other scheme over the same %d steps after %d warm-up steps, each "
                                            "behind a try/except and a %.0f s wall-clock guard; value = the fastest that finished" % (args.steps, args.warmup, args.scheme_timeout),
                                     "timed_first": first_name, "picked": best,
                                     "ms_per_step": {k: v["t"] / args.steps * 1e3 for k, v in done.items()}}
            if others:
                ex["alt"] = others[0]
            if len(others) > 1:
                ex["alt2"] = others[1]
            if len(others) > 2:
                ex["alt_more"] = others[2:]
        return build_line(head, ex)

    if multi and not args.no_alt:
        for name in [n for n in args.schemes.split(',') if n and n != first_name]:
            watchdog.arm(args.scheme_timeout, 'exchange scheme %s' % name,
                         lambda msg, name=name: emit(dict(final_line({"exchange": name, "error": msg}), watchdog_fired=True)))
            try:
                results[name] = run_scheme(make_comm(name))
            except Exception as e:                       # (other ranks may now be out of step: the guard of the next scheme covers that)
                failed.append({"exchange": name, "error": repr(e)})
                log('[bench] rank %d: exchange scheme %s failed: %r' % (rank, name, e))
            finally:
                watchdog.disarm()
    # (round 6, --graph) ... and the fastest capturable scheme once more with its step captured in a hipGraph and replayed (~60 launches per
    # 4-5 ms rank-step at 8 ranks): a candidate for `value` like any other scheme when it follows the eager run's losses (same Philox
    # stream, same number of steps: the last training loss must agree to 1e-3); behind the same guards
    if multi and args.graph and not staged:
        done_now = {k: v for k, v in results.items() if k is not None and make_comm(k).capturable}
        base = min(done_now, key=lambda e: done_now[e]["t"]) if done_now else None
        if base is not None:
            name = base + '+hipgraph'
            watchdog.arm(min(args.scheme_timeout, 180.0), 'exchange scheme %s' % name,
                         lambda msg, name=name: emit(dict(final_line({"exchange": name, "error": msg}), watchdog_fired=True)))
            try:
                r = run_scheme(make_comm(base), hip_graph=True)
                r["kern_ms"] = r["kern_ms"] or done_now[base]["kern_ms"]          # (replays record no per-kernel events: the eager run's)
                # (the captured run warms up for at least four steps: bring the eager model of the same scheme to the same step count first)
                last_e = done_now[base]["last"]
                for _ in range(max(args.warmup, 4) - args.warmup):
                    last_e = done_now[base]["clf"].f_train(X, y_tr, y_dev, A, tr, dev)
                l0, l1 = float(last_e[0]), float(r["last"][0])
                if not (abs(l1 - l0) <= 1e-3 * abs(l0)):
                    raise RuntimeError("captured run's last training loss %r differs from the eager run's %r" % (l1, l0))
                results[name] = r
            except Exception as e:
                failed.append({"exchange": name, "error": repr(e)})
                log('[bench] rank %d: exchange scheme %s failed: %r' % (rank, name, e))
            finally:
                watchdog.disarm()
    if multi and not args.no_check:
        watchdog.arm(args.scheme_timeout, 'partition_check',
                     lambda msg: emit(dict(final_line(), partition_check={"error": msg}, watchdog_fired=True)))
        try:
            comms = {k: v["comm"] for k, v in results.items() if k is not None and '+' not in k}
            extras["partition_check"] = partition_check(make_clf, comms, X, A, Y, tr, dev, rank)
        except Exception as e:
            extras["partition_check"] = {"error": repr(e)}
        finally:
            watchdog.disarm()

    if rank == 0:
        clf, g = first["clf"], first["g"]
        if world == 1:
            rf = {}
            if (args.traffic == 'live' and first["F"] == 300 and args.shape == 'twus' and args.reorder is None and not bf16_operand
                    and not overrides and not force_dist):
                # the dominant kernel's HBM-side traffic, counted now (the child runs the same job for 2 steps under the counters)
                kern = 'spmm_rows_kernel<%d, 0, 16, 0, 0>' % ((first["F"] + 63) // 64)
                child = ['--pmc-child', '--steps', '2', '--warmup', '1', '--shape', args.shape, '--dropout', str(args.dropout),
                         '--gemm-precision', precision, '--hid'] + [str(h) for h in args.hid]
                log('[bench] counting the HBM-side traffic of %s (two rocprofv3 --pmc child passes)...' % kern)
                torch.cuda.synchronize()
                live_traffic.update(traffic_live(child, kern))
                live_traffic["kernel"] = kern
                if live_traffic.get("error"):
                    log('[bench] ... not counted: %s' % live_traffic["error"])
            if args.shape != 'cmu' and not args.pmc_child:
                try:
                    rf["others"] = other_kernels(clf, lambda: clf.f_train(X, y_tr, y_dev, A, tr, dev), g, args.hid, N, C, precision)
                except Exception as e:                       # evidence only: never fail the headline line over it
                    rf["others_error"] = repr(e)
            try:
                if not args.pmc_child:
                    extras["clocks"] = ClockSampler(local).run(lambda: clf.f_train(X, y_tr, y_dev, A, tr, dev), 3.0, torch.cuda.synchronize)
            except Exception as e:
                extras["clocks"] = {"error": repr(e)}
            if not args.no_extras and args.shape != 'cmu':
                # what a gather of the product's row size can move on this GPU, measured now: the algorithmic fraction above
                # next to the fraction of THAT ceiling the counted traffic reaches
                try:
                    gc_ = gather_ceiling(512, 4 * ops.gather_ld(first["F"]))
                    if gc_ is not None:
                        rf["gather_ceiling"] = gc_
                        if gc_.get("tbps"):
                            rf["gather_ceiling_tbps"] = gc_["tbps"]
                            rf["gather_ceiling_source"] = "%s at commit %s" % (gc_.get("source"), _git_commit())
                            pm_traffic = build_line(first, {})["roofline"]["traffic"]
                            avg_ms = float(np.mean(first["kern_ms"])) if first["kern_ms"] else None
                            if pm_traffic and avg_ms:
                                rf["frac_of_gather_ceiling"] = pm_traffic / (avg_ms * 1e-3) / 1e12 / gc_["tbps"]
                                rf["frac_of_gather_ceiling_how"] = ("counted HBM-side traffic per launch (`traffic`) / this run's average launch time / "
                                                                    "gather_ceiling_tbps: how close the product is to what a gather can move; `frac` stays the "
                                                                    "SURVEY.md 8d algorithmic fraction")
                except Exception as e:
                    rf["gather_ceiling_error"] = repr(e)
                try:
                    extras["layer"] = layer_block(g, N, nnz, args.hid[0], device, precision)
                except Exception as e:
                    extras["layer"] = {"error": repr(e)}
            extras["roofline"] = rf
            if not args.no_extras and precision == 'bf16x3':
                # the same job with the exact fp32 MFMA in every product, same K / W, right after the headline (same box, same clocks state)
                try:
                    del clf
                    r32 = run_scheme(None, 'f32')
                    extras["alt_exact_f32"] = {"value": n_conv * nnz * args.steps / r32["t"], "unit": "edges/s", "ms_per_step": r32["t"] / args.steps * 1e3,
                                               "step_ms_median": r32["step_ms"][len(r32["step_ms"]) // 2], "train_loss_last": float(r32["last"][0]),
                                               "gemm": "exact fp32 MFMA (v_mfma_f32_16x16x4_f32) in every product (--gemm-precision f32)",
                                               "note": "same job, same K / W, timed after the headline in the same process"}
                    del r32
                except Exception as e:
                    extras["alt_exact_f32"] = {"error": repr(e)}
        if world == 1 and args.cpu_sample != 'none':
            log('[bench] timing the CPU oracle (%s sample)...' % args.cpu_sample)
            extras["cpu_baseline"] = cpu_baseline(args.shape, A, X, Y, tr, dev, args.hid, C, args.cpu_sample)
            log('[bench] ... and with the sparse products on all cores')
            try:
                extras["cpu_baseline_mt"] = cpu_baseline(args.shape, A, X, Y, tr, dev, args.hid, C, args.cpu_sample, multithreaded=True)
            except Exception as e:
                extras["cpu_baseline_mt"] = {"error": repr(e)}
        else:
            extras["cpu_baseline"] = None
        emit(final_line())
    if distributed:
        watchdog.arm(60.0, 'destroy_process_group', lambda msg: None)
        torch.distributed.destroy_process_group()
        watchdog.disarm()


if __name__ == '__main__':
    main()
