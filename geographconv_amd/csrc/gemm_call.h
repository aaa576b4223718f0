// Host-side description of one GEMM launch, shared by gemm.hip (exact fp32 kernels, dispatch) and gemm_x3.hip (the split-bf16 kernels).
#pragma once
#include "common.h"

#include <algorithm>

namespace geogcn {

// One launch multiplies up to two SEGMENTS (gcnmodel.py:281-286: the highway block's conv branch and gate read the same input; their
// backward adds two products into the same dH):
//   n_nseg = 2  "dual":            C[q] = act_q(op(A[0]) . B[q] + bias[q]), q = 0, 1
//   n_kseg = 2  "k-concatenated":  C[0] = A[0].op(B[0]) + A[1].op(B[1]) [+ C[0]] -- one accumulator, one pass over C
// Never both.
struct GemmCall {
    int64_t M;
    int n_nseg, n_kseg;
    const float* A[2]; int64_t lda[2];
    const float* B[2]; int64_t ldb[2];
    float* C[2]; int64_t ldc[2];
    const float* bias[2];
    int64_t N[2], K[2];
    int act[2];
    int accumulate;
    int panel_w = 0;
    int64_t panel_R = 0;
    const float* gateG = nullptr; int64_t ldg = 0; const float* gateT = nullptr; int64_t ldt = 0;      // whole-rows kernels only
    const float* postY = nullptr; int64_t ldy = 0; const uint8_t* postKeep = nullptr; int64_t postF = 0; float postScale = 0.f;
    int precision = GEOGCN_GEMM_F32;         // GEOGCN_GEMM_F32 (exact) or GEOGCN_GEMM_BF16X3 (fp32-class split-bf16 where a kernel takes the shape)
    int64_t maxN() const { return n_nseg == 2 ? std::max(N[0], N[1]) : N[0]; }
};

// ---- gemm_x3.hip: the fp32-class split-bf16 ("bf16x3") contraction on the whole-rows and A^T.B structures ----------------------------
// whole rows (A . B, A . B^T; dual, k-concatenated, gate-carry / tanh-gradient epilogues): 0 = shape not taken
int x3_rows_kc(const GemmCall& c, bool transA, bool transB);
size_t x3_rows_ws_bytes(const GemmCall& c, int kc);
int x3_run_rows(int kc, bool transB, const GemmCall& c, void* ws, hipStream_t st);
// A^T . B split-K slabs (the caller combines them with splitk_reduce_launch): which (bm, bn) tiles the kernel has
bool x3_tn_takes(int bm, int bn);
struct X3TnCall {
    int64_t M, K;
    const float* A; int64_t lda;
    const float* B[2]; int64_t ldb[2]; int64_t N[2];
    float* W; int64_t ldw, seg_w;       // slabs [nsplit][M][ldw]; N segment q starts at column q * seg_w
    int n_mt, n_nt, nt_per_seg, nsplit;
    int64_t kchunk;                     // a multiple of 32
};
int x3_tn_launch(int bm, int bn, const X3TnCall& t, hipStream_t st);

}  // namespace geogcn
