import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from geographconv_amd import ops, synth
from tools.bench_kernels import timeit
dev = torch.device('cuda:0')
A = synth.powerlaw_ahat(440000, 10000000)
dA = ops.CSR(A, dev)
for F in (32, 40, 64, 76, 80):
    Z = ops.DMat(440000, F, dev, ld=F if F % 4 == 0 else None)
    Z.t.normal_()
    out = ops.DMat(440000, F, dev)
    print('F=%d  %.3f ms' % (F, timeit(lambda: ops.spmm(dA, Z, out=out), 20)[0]), flush=True)
