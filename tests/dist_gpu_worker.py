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
    if dist.get_rank() == 0:
        print('DIST_GPU_OK world=%d' % dist.get_world_size())
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
