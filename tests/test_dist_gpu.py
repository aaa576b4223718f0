"""RCCL path on real hardware: torch.distributed.run + nccl backend + TorchDistComm + the HIP kernels.
A 1-GPU box can only launch world_size 1, which still exercises process-group setup, the in-place
all_gather_into_tensor on device buffers, the flat gradient all-reduce and the partition plumbing
(`_force_dist`); world_size 2 over gloo with the same code is covered on CPU (test_dist_cpu.py)."""
import os
import subprocess
import sys

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
