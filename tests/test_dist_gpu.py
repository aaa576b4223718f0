"""The partitioned path on real hardware.  RCCL (torch.distributed.run + nccl, and the library's own geogcn_comm_* entry
points) runs on however many GPUs are visible -- one on the test box, which still exercises process-group setup, the
in-place all_gather_into_tensor on device buffers, the flat gradient all-reduce and the partition plumbing (`_force_dist`).
The multi-rank logic WITH the HIP kernels is covered by letting 2 / 3 / 4 / 8 ranks share cuda:0 with the collectives staged
through the host and gloo (`GEOGCN_DIST_BACKEND=staged-gloo`, dist.HostStagedGloo); with the NumPy double under gloo it is covered on CPU
(test_dist_cpu.py)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_torchrun_nccl_world1_matches_golden():
    import torch
    n = min(2, torch.cuda.device_count())
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n),
           '--master-addr', '127.0.0.1', '--master-port', '29517', os.path.join(root, 'tests', 'dist_gpu_worker.py')]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert 'DIST_GPU_OK world=%d' % n in r.stdout
    assert 'captured_schemes=2' in r.stdout          # (round 6) the partitioned step under hipGraph over RCCL: the two all-gather schemes (all-to-all steps stay eager)


def test_torchrun_native_rccl_entry_points_match_golden():
    """The same worker with the data path on the library's own RCCL entry points (include/geogcn.h geogcn_comm_*,
    GEOGCN_DIST_BACKEND=native) instead of torch.distributed's."""
    import torch
    n = min(2, torch.cuda.device_count())
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', GEOGCN_DIST_BACKEND='native')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n),
           '--master-addr', '127.0.0.1', '--master-port', '29519', os.path.join(root, 'tests', 'dist_gpu_worker.py')]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert 'DIST_GPU_OK world=%d' % n in r.stdout
    assert 'backend=NativeRccl' in r.stdout and 'captured_schemes=2' in r.stdout


def test_comm_entry_points_world1():
    """geogcn_comm_* at world 1 (all a 1-GPU box allows): unique id, init, the three collectives on device buffers (an
    all-reduce over one rank is the identity, the all-gather is in place, the all-to-all copies the own panel), error
    codes, destroy."""
    import ctypes as C

    import torch
    from geographconv_amd import _ffi, dist as gdist
    lib = _ffi.lib()
    assert lib.geogcn_comm_available() == 1
    dev = torch.device('cuda:0')
    uid = gdist.NativeRccl.unique_id()
    assert len(uid) == _ffi.COMM_ID_BYTES and any(uid)
    small = C.create_string_buffer(16)
    assert lib.geogcn_comm_unique_id(small, 16) == -2                       # GEOGCN_E_SIZE
    c = gdist.NativeRccl(1, 0, uid, dev)
    assert lib.geogcn_comm_world(c._h) == 1 and lib.geogcn_comm_rank(c._h) == 0
    x = torch.arange(1000, dtype=torch.float32, device=dev)
    ref = x.clone()
    c.all_reduce(x)
    w = c.all_reduce(x, async_op=True)
    w.wait()
    torch.cuda.synchronize()
    assert torch.equal(x, ref)
    buf = torch.randn(64, 40, device=dev)
    keep = buf.clone()
    c.all_gather_into_tensor(buf, buf[0:64])
    send = torch.randn(3, 128, device=dev).to(torch.bfloat16)
    recv = torch.zeros_like(send)
    c.all_to_all_single(recv.view(-1), send.view(-1), async_op=True).wait()
    torch.cuda.synchronize()
    assert torch.equal(buf, keep) and torch.equal(recv, send)
    assert lib.geogcn_comm_alltoall(c._h, C.c_void_p(send.data_ptr()), C.c_void_p(send.data_ptr()), 16, None) == -1   # aliased
    # the halo exchange's per-peer sizes (row counts of pitched 2-D buffers -> geogcn_comm_alltoallv in bytes): at one rank
    # the only piece is the own one; zero-sized pieces are skipped
    rows = torch.randn(5, 64, device=dev)
    got = torch.zeros(5, 64, device=dev)
    c.all_to_all_single(got, rows, output_split_sizes=[5], input_split_sizes=[5], async_op=True).wait()
    c.all_to_all_single(got[:0], rows[:0], output_split_sizes=[0], input_split_sizes=[0])
    torch.cuda.synchronize()
    assert torch.equal(got, rows)
    one = (C.c_int64 * 1)(64)
    assert lib.geogcn_comm_alltoallv(c._h, C.c_void_p(rows.data_ptr()), one, C.c_void_p(rows.data_ptr()), one, None) == -1   # aliased
    assert lib.geogcn_comm_alltoallv(c._h, C.c_void_p(rows.data_ptr()), None, C.c_void_p(got.data_ptr()), one, None) == -1
    neg = (C.c_int64 * 1)(-4)
    assert lib.geogcn_comm_alltoallv(c._h, C.c_void_p(rows.data_ptr()), neg, C.c_void_p(got.data_ptr()), one, None) == -2
    assert lib.geogcn_comm_allreduce_sum_f32(None, C.c_void_p(x.data_ptr()), 4, None) == -1
    c.close()


@pytest.mark.parametrize("world", [2, 3, 4, 8])
def test_real_kernels_under_a_real_partition_on_one_gpu(world):
    """The combination no 1-GPU box can otherwise reach: the HIP kernels AND a multi-rank partition.  `world` processes
    share cuda:0; the collectives are staged through the host and gloo (dist.HostStagedGloo).  Same
    assertions as the RCCL worker: golden vectors through both exchange schemes, the partitioned graph product bitwise equal
    to the one-GPU kernel on the same rows, the bf16 configuration through both schemes."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GEOGCN_DIST_BACKEND='staged-gloo')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world),
           '--master-addr', '127.0.0.1', '--master-port', str(29521 + world), os.path.join(root, 'tests', 'dist_gpu_worker.py')]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert 'DIST_GPU_OK world=%d' % world in r.stdout


def _run_bench_self_launched(extra, timeout, env_extra=None):
    """Exactly the driver's command shape -- `python bench.py --gpus N ...`, NO torchrun, WORLD_SIZE unset -- with the ranks
    sharing cuda:0 and the collectives staged through the host (a functional check, never a measurement).  Rank 0 prints the first
    scheme's line as soon as it has it (marked "early line"), then THE line, which is the last line of stdout."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GEOGCN_DIST_BACKEND='staged-gloo', **(env_extra or {}))
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(root, 'bench.py')] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert 1 <= len(lines) <= 2, r.stdout[-2000:]             # rank 0 only: [the early line,] the line
    assert r.stdout.strip().splitlines()[-1] == lines[-1]     # and it is the LAST line of stdout
    for l in lines[:-1]:
        e = json.loads(l)
        assert e['note'].startswith('early line') and e['value'] > 0 and e['config']['dist']['exchange'] == 'allgather'
    return json.loads(lines[-1])


def test_bench_launches_its_own_ranks_and_checks_itself():
    """`python bench.py --gpus 2 --shape cmu --steps 3 --warmup 1` by itself: bench.py starts the two ranks (one process per
    GPU through torch.distributed.run), rank 0 prints ONE line carrying the world size it saw, both exchange schemes
    (`value` = the default, `alt` = the other) and a `partition_check` block: two dropout-free steps through every scheme
    against the same process's un-partitioned steps."""
    d = _run_bench_self_launched(['--gpus', '2', '--shape', 'cmu', '--steps', '3', '--warmup', '1'], 900)
    assert d['n_gpus'] == 2 and d['config']['world_size'] == 2 and d['config']['parallelism'] == 'rows2'
    assert d['config']['dist']['world_size_seen'] == 2 and d['config']['dist']['staged'] is True
    assert 'NOT a measurement' in d['config']['collectives'] and d['value'] > 0 and d['scaling'] == 'strong'
    assert d['roofline']['kernel'] and np.isfinite(d['config']['train_loss_last'])
    # `value` is the fastest of the schemes timed over the same K steps; the north_star's all-gather is timed first (round 5), every time
    # is in exchange_choice; who is here is in config.dist
    ch = d['exchange_choice']
    assert ch['timed_first'] == 'allgather' and set(ch['ms_per_step']) == {'allgather', 'a2a', 'agpipe'}
    assert [r['rank'] for r in d['config']['dist']['ranks']] == [0, 1] and d['config']['dist']['rccl_version']
    assert d['config']['dist']['exchange'] == ch['picked'] == min(ch['ms_per_step'], key=ch['ms_per_step'].get)
    assert abs(d['ms_per_step'] - ch['ms_per_step'][ch['picked']]) < 1e-9
    others = {d['alt']['exchange'], d['alt2']['exchange']}
    assert others == {'allgather', 'a2a', 'agpipe'} - {ch['picked']} and d['alt']['value'] > 0 and d['alt2']['value'] > 0
    pc = d['partition_check']
    for scheme in ('allgather', 'a2a', 'agpipe'):
        c = pc[scheme]
        assert c['max_abs_dloss'] <= 2e-5 and c['max_abs_dacc'] <= 2e-4, (scheme, c)      # (one near-tie row of 5,685 at most)
        assert c['max_abs_dP'] <= 2e-6 and c['argmax_agreement'] >= 0.9995, (scheme, c)


def test_bench_whole_multi_gpu_flow_over_rccl_at_world_1():
    """(round 6) The N > 1 flow of bench.py -- early line, every exchange scheme behind its guard, the hipGraph variant of the fastest
    capturable scheme (--graph), the partition check -- over REAL RCCL (torch.distributed nccl, no host staging) on the one GPU the
    box has: GEOGCN_BENCH_FORCE_DIST=1 runs the partitioned code path at world 1.  One rank's collectives are trivial, but the process
    group, the in-place all-gathers on device buffers, the all-to-alls, the gradient all-reduce and the capture all run on the library
    the driver's 8-GPU run will use."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GEOGCN_BENCH_FORCE_DIST='1', HSA_ENABLE_IPC_MODE_LEGACY='0', MASTER_PORT='29561')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'GEOGCN_DIST_BACKEND'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(root, 'bench.py'), '--shape', 'cmu', '--steps', '3', '--warmup', '2', '--no-extras', '--traffic', 'none',
           '--cpu-sample', 'none', '--graph']
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 2 and json.loads(lines[0])['note'].startswith('early line')
    d = json.loads(lines[-1])
    assert d['n_gpus'] == 1 and d['config']['dist']['torch_backend'] == 'nccl' and d['config']['dist']['staged'] is False
    ch = d['exchange_choice']
    assert ch['timed_first'] == 'allgather' and {'allgather', 'a2a', 'agpipe'} <= set(ch['ms_per_step'])
    graphed = [k for k in ch['ms_per_step'] if k.endswith('+hipgraph')]
    assert len(graphed) == 1 and graphed[0].split('+')[0] in ('allgather', 'agpipe'), ch          # (all-to-all steps are not captured)
    assert not any('error' in x for x in (d.get('alt'), d.get('alt2')) + tuple(d.get('alt_more', ())) if x), d
    assert d['config']['hip_graph'] == (ch['picked'] in graphed)
    for scheme in ('allgather', 'a2a', 'agpipe'):
        c = d['partition_check'][scheme]
        assert c['max_abs_dloss'] <= 2e-5 and c['max_abs_dP'] <= 2e-6 and c['argmax_agreement'] >= 0.9995, (scheme, c)


def test_bench_survives_schemes_that_fail():
    """One exception in a scheme timed after the first must not cost the headline (the first contact with real links
    may be the only one): with a failure injected into a2a AND agpipe on every rank, the last line of stdout is still one parsable
    JSON line whose `value` is the all-gather's, and the failed schemes are reported as errors."""
    d = _run_bench_self_launched(['--gpus', '2', '--shape', 'cmu', '--steps', '2', '--warmup', '1', '--no-check'], 900,
                                 {'GEOGCN_BENCH_INJECT': 'a2a:raise;agpipe:raise'})
    assert d['value'] > 0 and d['config']['dist']['exchange'] == 'allgather' and d['exchange_choice']['picked'] == 'allgather'
    errs = {d['alt']['exchange']: d['alt']['error'], d['alt2']['exchange']: d['alt2']['error']}
    assert set(errs) == {'a2a', 'agpipe'} and all('injected failure' in e for e in errs.values())


def test_bench_survives_a_scheme_that_hangs():
    """... and a scheme that never returns (a collective stuck on a link) is cut off by its wall-clock guard: rank 0 prints the line it
    has -- the schemes that finished, the stuck one as a timeout -- and every rank leaves."""
    d = _run_bench_self_launched(['--gpus', '2', '--shape', 'cmu', '--steps', '2', '--warmup', '1', '--no-check', '--scheme-timeout', '12'], 900,
                                 {'GEOGCN_BENCH_INJECT': 'agpipe:hang'})
    assert d['value'] > 0 and set(d['exchange_choice']['ms_per_step']) == {'allgather', 'a2a'}
    stuck = [x for x in (d.get('alt'), d.get('alt2')) if x and x['exchange'] == 'agpipe']
    assert stuck and 'timeout' in stuck[0]['error'] and d['watchdog_fired'] is True


def test_bench_partitioned_at_the_full_twitterus_shape():
    """BASELINE configs[3] at FULL size (N = 440,000, 10.7 M stored edges, 3 x 300) through the same command: two ranks
    with the real kernels, both exchange schemes, each compared with rank 0's un-partitioned steps (what
    tools/staged_twus_check.py prints; measured there: losses to 1e-6, max |dP| 3e-9, argmax agreement >= 0.999995)."""
    d = _run_bench_self_launched(['--gpus', '2', '--steps', '1', '--warmup', '1'], 2400)
    assert d['config']['world_size'] == 2 and 'N=440000' in d['config']['workload']
    pc = d['partition_check']
    for scheme in ('allgather', 'a2a', 'agpipe'):
        c = pc[scheme]
        assert c['max_abs_dloss'] <= 1e-5 and c['max_abs_dacc'] <= 4e-6, (scheme, c)      # (one near-tie row of 264,000 at most)
        assert c['max_abs_dP'] <= 5e-8 and c['argmax_agreement'] >= 0.99999, (scheme, c)


def test_bench_partitioned_config5_shape_bf16_full_size():
    """BASELINE configs[4]'s model (six 600-wide highway layers, bf16 H.W products, bf16 gathered operand AND bf16 on the wire)
    at the full TwitterUS size through the same self-launching command: two ranks, both exchange schemes, each against rank
    0's un-partitioned bf16 steps.  Measured (profiles/r03_l_staged_cfg5_w2.json): losses to 1.4e-6, identical accuracies, max
    |dP| 1.9e-7, argmax agreement >= 0.999975 (the partitioned GEMMs tile 220,000 rows per rank: the fp32 accumulation order
    inside a product is unchanged, the loss sums are added over two partial sums)."""
    d = _run_bench_self_launched(['--gpus', '2', '--hid', '600', '600', '600', '600', '600', '600', '--gemm-precision', 'bf16',
                                  '--steps', '1', '--warmup', '1', '--schemes', 'a2a'], 2400)
    assert d['config']['world_size'] == 2 and d['dtype'] == 'bf16' and '600x600x600x600x600x600' in d['config']['workload']
    # (round 6: the north_star's all-gather and the feature repartition -- the scheme with bf16 PANELS on the wire; the slab-pipelined
    #  all-gather at this size is covered by the 3 x 300 test above and, in bf16, at small sizes by tests/dist_gpu_worker.py: -45 s)
    for scheme in ('allgather', 'a2a'):
        c = d['partition_check'][scheme]
        assert c['max_abs_dloss'] <= 1e-5 and c['max_abs_dacc'] <= 4e-6, (scheme, c)
        assert c['max_abs_dP'] <= 2e-6 and c['argmax_agreement'] >= 0.9999, (scheme, c)


def test_gcnmain_row_partitioned_on_one_gpu(tmp_path):
    """The reference's entry point under torch.distributed.run with 3 ranks (sharing cuda:0, staged collectives): fit with
    early stopping decided identically on every rank, predict + geo_eval through the row partition -- same dev accuracy and
    distances as the single-process run with the same flags (p = 0: no dropout stream to keep in step)."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    flags = '-hid 48 48 -reg 0.0 -dropout 0.0 -highway -silent --synthetic cmu --epochs 6 -maxdown 2'.split()
    probe = ("import sys, json; sys.path.insert(0, %r); from geographconv_amd import gcnmain; "
             "clf, res = gcnmain.run(sys.argv[1:]); "
             "import os; print('RESULT ' + json.dumps(res[0]['dev'])) if int(os.environ.get('RANK', '0')) == 0 else None") % root
    env1 = dict(os.environ)
    env1.pop('WORLD_SIZE', None)
    r1 = subprocess.run([sys.executable, '-c', probe] + flags, capture_output=True, text=True, timeout=900, env=env1, cwd=str(tmp_path))
    assert r1.returncode == 0, r1.stdout[-2000:] + r1.stderr[-3000:]
    single = json.loads([l for l in r1.stdout.splitlines() if l.startswith('RESULT ')][0][7:])
    env = dict(os.environ, GEOGCN_DIST_BACKEND='staged-gloo')
    (tmp_path / 'probe.py').write_text(probe)                  # (torch.distributed.run wants a script path)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '3', '--master-addr', '127.0.0.1',
           '--master-port', '29573', str(tmp_path / 'probe.py')] + flags
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    part = json.loads([l for l in r.stdout.splitlines() if l.startswith('RESULT ')][0][7:])
    assert abs(part[2] - single[2]) <= 0.2, (part, single)                      # dev acc@161 (%)
    assert abs(part[0] - single[0]) <= 0.02 * max(1.0, single[0]) and abs(part[1] - single[1]) <= 0.05 * max(1.0, single[1]), (part, single)
