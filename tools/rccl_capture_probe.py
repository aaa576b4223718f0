"""Does a hipGraph capture survive RCCL collectives on this box?  (round 6; world 1 is all a one-GPU box allows)
   python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29555 tools/rccl_capture_probe.py [op ...]
Each op is tried in its own process by the caller (a crash inside librccl takes the process with it)."""
import faulthandler
import os
import sys

import torch
import torch.distributed as dist

faulthandler.enable()


def main():
    ops = sys.argv[1:] or ['allreduce']
    dev = torch.device('cuda', int(os.environ.get('LOCAL_RANK', '0')))
    torch.cuda.set_device(dev)
    dist.init_process_group('nccl', device_id=dev)
    w = dist.get_world_size()
    x = torch.ones(1 << 20, device=dev)
    big = torch.zeros(w * (1 << 20), device=dev)
    y = torch.zeros(1 << 20, device=dev)
    # eager first (communicator fully set up)
    dist.all_reduce(x)
    dist.all_gather_into_tensor(big, x)
    dist.all_to_all_single(y, x)
    torch.cuda.synchronize()
    print('eager ok', flush=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        x.mul_(0.5)
        for op in ops:
            if op == 'allreduce':
                dist.all_reduce(x)
            elif op == 'allreduce_async':
                dist.all_reduce(x, async_op=True).wait()
            elif op == 'allgather':
                dist.all_gather_into_tensor(big, x)
            elif op == 'allgather_async':
                dist.all_gather_into_tensor(big, x, async_op=True).wait()
            elif op == 'a2a':
                dist.all_to_all_single(y, x)
            elif op == 'a2a_async':
                dist.all_to_all_single(y, x, async_op=True).wait()
            elif op == 'none':
                pass
        x.add_(1.0)
    print('captured', ops, flush=True)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    print('replayed', ops, float(x[0]), flush=True)
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
