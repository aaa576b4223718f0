"""Does the fabric-bound graph product overlap with an MFMA-bound GEMM when the two are launched on different HIP streams?
(In the highway block the gate's products depend on H / dU only, never on the SpMM of the same layer.)
   python tools/overlap_probe.py [--reps 10]
Prints: each kernel alone, the two back to back on one stream, and the two on two streams (both launch orders)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from geographconv_amd import ops, synth  # noqa: E402


def wall(fn, reps, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    out = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        b.synchronize()
        out.append(a.elapsed_time(b))
    out.sort()
    return out[len(out) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=10)
    ap.add_argument('--F', type=int, default=300)
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    s = synth.SHAPES['twus']
    A = synth.powerlaw_ahat(s.N, s.E_target)
    dA = ops.CSR(A, dev)
    F = args.F
    Z = ops.DMat.empty(s.N, F, dev, ld=ops.gather_ld(F)); Z.t.normal_()
    S = ops.DMat(s.N, F, dev)
    H = ops.DMat.empty(s.N, F, dev); H.t.normal_()
    W = ops.DMat.empty(F, F, dev); W.t.normal_(std=0.05)
    b = torch.zeros(F, device=dev)
    T = ops.DMat(s.N, F, dev)
    dW = ops.DMat(F, F, dev)
    side = torch.cuda.Stream()
    main_s = torch.cuda.current_stream()

    def spmm():
        ops.spmm(dA, Z, out=S, bias=b, act=ops.ACT_TANH)

    def gemm_nn():
        ops.gemm(H, W, out=T, bias=b, act=ops.ACT_SIGMOID)

    def gemm_nt():
        ops.gemm(H, W, out=T, transB=True)

    def gemm_tn():
        ops.gemm(H, Z, out=dW, transA=True)

    def two_streams(first, second):
        """`first` on the main stream, `second` on the side stream, both started at the same point, joined at the end."""
        def run():
            ev = torch.cuda.Event()
            ev.record(main_s)
            side.wait_event(ev)
            with torch.cuda.stream(side):
                second()
                done = torch.cuda.Event()
                done.record(side)
            first()
            main_s.wait_event(done)
        return run

    t_spmm = wall(spmm, args.reps)
    print('spmm alone                 %.3f ms' % t_spmm)
    for name, g in (('gemm NN sigmoid', gemm_nn), ('gemm NT', gemm_nt), ('gemm TN', gemm_tn)):
        t_g = wall(g, args.reps)
        t_seq = wall(lambda: (spmm(), g()), args.reps)
        t_a = wall(two_streams(spmm, g), args.reps)
        t_b = wall(two_streams(g, spmm), args.reps)
        print('%-16s alone %.3f   one stream %.3f   two streams: spmm on main %.3f, gemm on main %.3f   (saved %.3f of %.3f)'
              % (name, t_g, t_seq, t_a, t_b, t_seq - min(t_a, t_b), t_g))
    g2 = lambda: (gemm_nt(), gemm_tn())
    t_g = wall(g2, args.reps)
    t_seq = wall(lambda: (spmm(), g2()), args.reps)
    t_a = wall(two_streams(spmm, g2), args.reps)
    print('NT + TN          alone %.3f   one stream %.3f   two streams %.3f   (saved %.3f)' % (t_g, t_seq, t_a, t_seq - t_a))


if __name__ == '__main__':
    main()
