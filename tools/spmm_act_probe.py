import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from geographconv_amd import ops, synth
dev = torch.device('cuda:0')
s = synth.SHAPES['twus']
A = synth.powerlaw_ahat(s.N, s.E_target)
dA = ops.CSR(A, dev)
rng = np.random.RandomState(1)
def t(fn, reps=7):
    for _ in range(2): fn()
    torch.cuda.synchronize(); ev=[]
    for _ in range(reps):
        a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True); a.record(); fn(); b.record(); ev.append((a,b))
    torch.cuda.synchronize(); ts=sorted(a.elapsed_time(b) for a,b in ev); return ts[len(ts)//2]
for F, bf in ((300,0),(600,1),(600,0),(300,1)):
    H = ops.DMat.empty(s.N, F, dev, ld=ops.gather_ld(F)); H.t.zero_()
    H.t[:, :F].copy_(torch.from_numpy((rng.randn(s.N, F)*0.3).astype(np.float32)))
    Hb = ops.cast_bf16(H) if bf else H
    out = ops.DMat(s.N, F, dev)
    bias = torch.from_numpy(np.pad(rng.randn(F).astype(np.float32), (0, ops.pad4(F)-F))).to(dev)
    r = {}
    r['plain'] = t(lambda: ops.spmm(dA, Hb, out=out))
    r['bias'] = t(lambda: ops.spmm(dA, Hb, out=out, bias=bias))
    r['tanh'] = t(lambda: ops.spmm(dA, Hb, out=out, act=ops.ACT_TANH))
    r['bias+tanh'] = t(lambda: ops.spmm(dA, Hb, out=out, bias=bias, act=ops.ACT_TANH))
    r['bias+sigmoid'] = t(lambda: ops.spmm(dA, Hb, out=out, bias=bias, act=ops.ACT_SIGMOID))
    print('F=%d bf16=%d' % (F, bf), {k: round(v,3) for k,v in r.items()}, flush=True)
