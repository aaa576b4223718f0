"""Long-row chunk placement A/B on one box: the plain and the highway-fused graph product (F = 300) with the chunks dealt round
the XCDs (chunks_with_owner = 0) and on the owning XCD (1), pinned TwitterUS-shape graph and the community graph under the
label-propagation order.   python tools/spmm_chunk_placement.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from geographconv_amd import graph, ops, synth  # noqa: E402
from tools.bench_kernels import timeit  # noqa: E402

dev = torch.device('cuda:0')
s = synth.SHAPES['twus']
F = 300
for name in ('pinned', 'sbm+lpa'):
    A = synth.powerlaw_ahat(s.N, s.E_target) if name == 'pinned' else synth.community_ahat(s.N, s.E_target, synth.TWUS_SBM_COMMUNITIES)
    if name != 'pinned':
        A = graph.reordering(A, 'lpa').matrix(A)
    Z = ops.DMat.empty(s.N, F, dev, ld=ops.gather_ld(F))
    Z.t.normal_()
    T, H = ops.DMat(s.N, F, dev), ops.DMat(s.N, F, dev)
    T.t.uniform_()
    H.t.normal_()
    b = torch.zeros(F, device=dev)
    out, Hc, Ho = ops.DMat(s.N, F, dev), ops.DMat(s.N, F, dev), ops.DMat(s.N, F, dev)
    auto = ops.CSR(A, dev).chunks_with_owner
    for local in (False, True, False, True):
        dA = ops.CSR(A, dev, local=local)
        p = timeit(lambda: ops.spmm(dA, Z, out=out), 20)[0]
        h = timeit(lambda: ops.spmm_highway(dA, Z, b, T, H, Hc=Hc, Hout=Ho), 20)[0]
        print('%-8s chunks_with_owner=%d (auto picks %d)  plain %.3f ms   highway-fused %.3f ms   [%d long rows, %d chunks]'
              % (name, local, auto, p, h, dA.n_long_rows, dA.n_chunks), flush=True)
        del dA
