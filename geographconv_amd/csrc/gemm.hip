// fp32 MFMA GEMM for the tall-skinny dense contractions of the GCN hot path (gfx950).
//   reference: T.dot(input, W) gcnmodel.py:126,149; lasagne DenseLayer gate gcnmodel.py:285;
//   and the Gemm ops Theano autodiff derives (dW = H^T.dZ, dH = dZ.W^T).
//
// v_mfma_f32_16x16x4_f32 (exact fp32, == fmaf chain; 157 TF peak).  Block = 4 waves (2x2),
// block tile BM x BN, K step 32, LDS double-buffered, register-staged global loads issued one
// tile ahead.  An operand is staged in one of two LDS images depending on how it lies in memory:
//   k-contiguous (A not transposed / B transposed): [rows][32+4], fragments read as float4 --
//       one ds_read_b128 feeds four MFMAs (k = kk + 4*(lane>>4) + t, t = 0..3);
//   k-strided    (A transposed / B not transposed): [32][cols+4], fragments read as 4 b32 with
//       the SAME k numbering, so any pairing of the two images multiplies matching k's.
// transA (reduction over the long dimension N_nodes) runs split-K into a workspace and a second
// kernel adds the slabs in slab order: deterministic, no atomics.
#include "common.h"

#include <algorithm>

namespace geogcn {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 32;
constexpr int KPITCH = BK + 4;     // k-contiguous image pitch (floats): 144 B rows, 16-B aligned
constexpr int TPB = 256;

template <int R>
struct KContig {                    // R rows x BK floats, row-major, pitch KPITCH
    static constexpr int kFloats = R * KPITCH;
    static constexpr int kIters = R / 32;          // 256 threads x float4 = 32 rows per pass
};
template <int Ccols>
struct KStrided {                   // BK rows x Ccols floats, pitch Ccols + 4
    static constexpr int kPitch = Ccols + 4;
    static constexpr int kFloats = BK * kPitch;
    static constexpr int kF4PerRow = Ccols / 4;
    static constexpr int kIters = (BK * kF4PerRow) / TPB;   // Ccols multiple of 32 => exact
};

__device__ __forceinline__ float4 load4_guard(const float* __restrict__ p, int64_t idx, int64_t lim) {
    // loads p[idx..idx+3] with elements >= lim replaced by 0 (p + idx is 16-B aligned)
    if (idx + 3 < lim) return *reinterpret_cast<const float4*>(p + idx);
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (idx < lim) r.x = p[idx];
    if (idx + 1 < lim) r.y = p[idx + 1];
    if (idx + 2 < lim) r.z = p[idx + 2];
    return r;
}

// ---- global -> registers ----------------------------------------------------------------------
// k-contiguous operand: memory [R_total][K] row-major (ld), tile rows r0.., k range k0..k0+31
template <int R>
__device__ __forceinline__ void gload_kcontig(float4 (&reg)[KContig<R>::kIters], const float* __restrict__ P,
                                              int64_t ld, int64_t r0, int64_t Rtot, int64_t k0, int64_t Kend) {
    const int tid = threadIdx.x;
    const int f4 = tid & 7;
    const int rr = tid >> 3;
#pragma unroll
    for (int i = 0; i < KContig<R>::kIters; ++i) {
        const int64_t row = r0 + rr + 32 * i;
        if (row < Rtot) reg[i] = load4_guard(P + row * ld, k0 + f4 * 4, Kend);
        else reg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}
template <int R>
__device__ __forceinline__ void sstore_kcontig(float* __restrict__ S, const float4 (&reg)[KContig<R>::kIters]) {
    const int tid = threadIdx.x;
    const int f4 = tid & 7;
    const int rr = tid >> 3;
#pragma unroll
    for (int i = 0; i < KContig<R>::kIters; ++i)
        *reinterpret_cast<float4*>(S + (rr + 32 * i) * KPITCH + f4 * 4) = reg[i];
}
// k-strided operand: memory [K][C_total] row-major (ld), tile cols c0.., k range k0..k0+31
template <int Ccols>
__device__ __forceinline__ void gload_kstrided(float4 (&reg)[KStrided<Ccols>::kIters], const float* __restrict__ P,
                                               int64_t ld, int64_t c0, int64_t Ctot, int64_t k0, int64_t Kend) {
    const int tid = threadIdx.x;
    constexpr int F4R = KStrided<Ccols>::kF4PerRow;
#pragma unroll
    for (int i = 0; i < KStrided<Ccols>::kIters; ++i) {
        const int e = tid + TPB * i;
        const int kr = e / F4R;
        const int c4 = e % F4R;
        const int64_t k = k0 + kr;
        if (k < Kend) reg[i] = load4_guard(P + k * ld, c0 + c4 * 4, Ctot);
        else reg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}
template <int Ccols>
__device__ __forceinline__ void sstore_kstrided(float* __restrict__ S, const float4 (&reg)[KStrided<Ccols>::kIters]) {
    const int tid = threadIdx.x;
    constexpr int F4R = KStrided<Ccols>::kF4PerRow;
#pragma unroll
    for (int i = 0; i < KStrided<Ccols>::kIters; ++i) {
        const int e = tid + TPB * i;
        const int kr = e / F4R;
        const int c4 = e % F4R;
        *reinterpret_cast<float4*>(S + kr * KStrided<Ccols>::kPitch + c4 * 4) = reg[i];
    }
}

template <int BM, int BN, bool AT, bool BT>
struct GemmCfg {
    static constexpr int kAFloats = AT ? KStrided<BM>::kFloats : KContig<BM>::kFloats;
    static constexpr int kBFloats = BT ? KContig<BN>::kFloats : KStrided<BN>::kFloats;
    static constexpr int kAIters = AT ? KStrided<BM>::kIters : KContig<BM>::kIters;
    static constexpr int kBIters = BT ? KContig<BN>::kIters : KStrided<BN>::kIters;
    static constexpr int kStageFloats = kAFloats + kBFloats;
    static constexpr size_t kLdsBytes = 2 * (size_t)kStageFloats * sizeof(float);
    static constexpr int MR = BM / 32;   // 16x16 tiles per wave along M (wave tile = BM/2)
    static constexpr int NR = BN / 32;
};

// MODE 0: C = act(acc + bias) [+ C if accumulate];  MODE 1: split-K slab (raw partial sums)
template <int BM, int BN, bool AT, bool BT, int ACT, int MODE>
__global__ __launch_bounds__(TPB) void gemm_kernel(int64_t M, int64_t N, int64_t K,
                                                   const float* __restrict__ A, int64_t lda,
                                                   const float* __restrict__ B, int64_t ldb,
                                                   float* __restrict__ C, int64_t ldc,
                                                   const float* __restrict__ bias, int accumulate,
                                                   int64_t kchunk, int n_mt, int n_nt, int xcd_order) {
    using Cfg = GemmCfg<BM, BN, AT, BT>;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    // XCD-aware tile order: consecutive blocks on one XCD (b, b+8, ...) walk the N tiles of one
    // M tile so the A panel is re-read from that XCD's L2.
    // (few M tiles -- the split-K case -- use the plain order so that every XCD gets work.)
    const int b = blockIdx.x;
    int mt, nt;
    if (xcd_order) {
        const int xcd = b % kNumXCD;
        const int j = b / kNumXCD;
        mt = (j / n_nt) * kNumXCD + xcd;
        nt = j % n_nt;
    } else {
        mt = b / n_nt;
        nt = b % n_nt;
    }
    if (mt >= n_mt) return;
    const int64_t m0 = (int64_t)mt * BM;
    const int64_t n0 = (int64_t)nt * BN;
    const int64_t kbeg = (int64_t)blockIdx.y * kchunk;
    const int64_t kend = min(K, kbeg + kchunk);

    const int lane = threadIdx.x & 63;
    const int wid = threadIdx.x >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int li = lane & 15, lg = lane >> 4;

    f32x4 acc[Cfg::MR][Cfg::NR];
#pragma unroll
    for (int i = 0; i < Cfg::MR; ++i)
#pragma unroll
        for (int jn = 0; jn < Cfg::NR; ++jn) acc[i][jn] = f32x4{0.f, 0.f, 0.f, 0.f};

    float4 ra[Cfg::kAIters], rb[Cfg::kBIters];
    auto gload = [&](int64_t k0) {
        if constexpr (AT) gload_kstrided<BM>(ra, A, lda, m0, M, k0, kend);
        else gload_kcontig<BM>(ra, A, lda, m0, M, k0, kend);
        if constexpr (BT) gload_kcontig<BN>(rb, B, ldb, n0, N, k0, kend);
        else gload_kstrided<BN>(rb, B, ldb, n0, N, k0, kend);
    };
    auto sstore = [&](int buf) {
        float* As = smem + buf * Cfg::kStageFloats;
        float* Bs = As + Cfg::kAFloats;
        if constexpr (AT) sstore_kstrided<BM>(As, ra);
        else sstore_kcontig<BM>(As, ra);
        if constexpr (BT) sstore_kcontig<BN>(Bs, rb);
        else sstore_kstrided<BN>(Bs, rb);
    };

    const int64_t nk = (kend > kbeg) ? (kend - kbeg + BK - 1) / BK : 0;
    if (nk > 0) {
        gload(kbeg);
        sstore(0);
    }
    __syncthreads();
    int cur = 0;
    for (int64_t kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) gload(kbeg + (kt + 1) * BK);
        const float* As = smem + cur * Cfg::kStageFloats;
        const float* Bs = As + Cfg::kAFloats;
#pragma unroll
        for (int kk = 0; kk < BK; kk += 16) {
            float af[Cfg::MR][4], bf[Cfg::NR][4];
#pragma unroll
            for (int i = 0; i < Cfg::MR; ++i) {
                const int mrow = wm * (BM / 2) + i * 16 + li;
                if constexpr (AT) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) af[i][t] = As[(kk + 4 * lg + t) * KStrided<BM>::kPitch + mrow];
                } else {
                    const float4 v = *reinterpret_cast<const float4*>(As + mrow * KPITCH + kk + 4 * lg);
                    af[i][0] = v.x; af[i][1] = v.y; af[i][2] = v.z; af[i][3] = v.w;
                }
            }
#pragma unroll
            for (int jn = 0; jn < Cfg::NR; ++jn) {
                const int ncol = wn * (BN / 2) + jn * 16 + li;
                if constexpr (BT) {
                    const float4 v = *reinterpret_cast<const float4*>(Bs + ncol * KPITCH + kk + 4 * lg);
                    bf[jn][0] = v.x; bf[jn][1] = v.y; bf[jn][2] = v.z; bf[jn][3] = v.w;
                } else {
#pragma unroll
                    for (int t = 0; t < 4; ++t) bf[jn][t] = Bs[(kk + 4 * lg + t) * KStrided<BN>::kPitch + ncol];
                }
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < Cfg::MR; ++i)
#pragma unroll
                    for (int jn = 0; jn < Cfg::NR; ++jn)
                        acc[i][jn] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][t], bf[jn][t], acc[i][jn], 0, 0, 0);
        }
        if (kt + 1 < nk) sstore(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // epilogue: C/D layout of the 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + r
    float* Cout = C;
    if constexpr (MODE == 1) Cout = C + (int64_t)blockIdx.y * M * ldc;
#pragma unroll
    for (int i = 0; i < Cfg::MR; ++i) {
#pragma unroll
        for (int jn = 0; jn < Cfg::NR; ++jn) {
            const int64_t col = n0 + wn * (BN / 2) + jn * 16 + li;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t row = m0 + wm * (BM / 2) + i * 16 + lg * 4 + r;
                if (row < M && col < N) {
                    float x = acc[i][jn][r];
                    if constexpr (MODE == 0) {
                        if (bias) x += bias[col];
                        x = apply_act<ACT>(x);
                        if (accumulate) x += Cout[row * ldc + col];
                    }
                    Cout[row * ldc + col] = x;
                }
            }
        }
    }
}

// split-K combine: C = act(sum_z slab[z] + bias) [+ C]
template <int ACT>
__global__ __launch_bounds__(TPB) void splitk_reduce_kernel(int64_t M, int64_t N, int nsplit,
                                                            const float* __restrict__ W, int64_t ldw,
                                                            float* __restrict__ C, int64_t ldc,
                                                            const float* __restrict__ bias, int accumulate) {
    const int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (e >= M * N) return;
    const int64_t row = e / N, col = e % N;
    float acc = 0.f;
    for (int z = 0; z < nsplit; ++z) acc += W[((int64_t)z * M + row) * ldw + col];
    if (bias) acc += bias[col];
    acc = apply_act<ACT>(acc);
    if (accumulate) acc += C[row * ldc + col];
    C[row * ldc + col] = acc;
}

struct SplitPlan {
    int nsplit;
    int64_t kchunk;
};

template <int BM, int BN>
SplitPlan plan_split(int64_t M, int64_t N, int64_t K) {
    const int64_t tiles = cdiv(M, BM) * cdiv(N, BN);
    int64_t ns = cdiv(3 * kNumCU, tiles);                 // ~3 blocks per CU
    const int64_t max_ns = std::max<int64_t>(1, K / (BK * 16));
    ns = std::max<int64_t>(1, std::min(ns, max_ns));
    int64_t kchunk = cdiv(cdiv(K, ns), BK) * BK;
    ns = cdiv(K, kchunk);
    return {(int)ns, kchunk};
}

template <int BM, int BN, bool AT, bool BT>
int launch_gemm(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B, int64_t ldb,
                float* C, int64_t ldc, const float* bias, int act, int accumulate, void* ws, size_t ws_bytes,
                hipStream_t st) {
    using Cfg = GemmCfg<BM, BN, AT, BT>;
    const int n_mt = (int)cdiv(M, BM), n_nt = (int)cdiv(N, BN);
    const int xcd_order = (n_mt >= 4 * kNumXCD) ? 1 : 0;
    const unsigned gx = xcd_order ? (unsigned)(cdiv(n_mt, kNumXCD) * kNumXCD * n_nt) : (unsigned)(n_mt * n_nt);
#define GEOGCN_GEMM_LAUNCH(ACT, MODE, grid, Cptr, ldC, kch)                                              \
    do {                                                                                                  \
        auto kern = gemm_kernel<BM, BN, AT, BT, ACT, MODE>;                                               \
        static bool attr_done = false;                                                                    \
        if (!attr_done) {                                                                                 \
            GEOGCN_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                           (int)Cfg::kLdsBytes));                                         \
            attr_done = true;                                                                             \
        }                                                                                                 \
        hipLaunchKernelGGL(kern, grid, dim3(TPB), Cfg::kLdsBytes, st, M, N, K, A, lda, B, ldb, Cptr, ldC, \
                           bias, accumulate, kch, n_mt, n_nt, xcd_order);                                          \
        GEOGCN_LAUNCH_CHECK("gemm_kernel");                                                               \
    } while (0)

    if (!AT) {
        const dim3 grid(gx, 1);
        const int64_t kch = cdiv(K, BK) * BK;
        if (act == GEOGCN_ACT_TANH) GEOGCN_GEMM_LAUNCH(GEOGCN_ACT_TANH, 0, grid, C, ldc, kch);
        else if (act == GEOGCN_ACT_SIGMOID) GEOGCN_GEMM_LAUNCH(GEOGCN_ACT_SIGMOID, 0, grid, C, ldc, kch);
        else GEOGCN_GEMM_LAUNCH(GEOGCN_ACT_NONE, 0, grid, C, ldc, kch);
        return 0;
    }
    // transA: split-K over the long reduction
    const SplitPlan sp = plan_split<BM, BN>(M, N, K);
    if (sp.nsplit == 1) {
        const dim3 grid(gx, 1);
        if (act == GEOGCN_ACT_TANH) GEOGCN_GEMM_LAUNCH(GEOGCN_ACT_TANH, 0, grid, C, ldc, sp.kchunk);
        else if (act == GEOGCN_ACT_SIGMOID) GEOGCN_GEMM_LAUNCH(GEOGCN_ACT_SIGMOID, 0, grid, C, ldc, sp.kchunk);
        else GEOGCN_GEMM_LAUNCH(GEOGCN_ACT_NONE, 0, grid, C, ldc, sp.kchunk);
        return 0;
    }
    const int64_t ldw = N;
    const size_t need = (size_t)sp.nsplit * (size_t)M * (size_t)ldw * sizeof(float);
    GEOGCN_REQUIRE(ws && ws_bytes >= need, GEOGCN_E_ARG, "gemm_f32: split-K workspace too small (%zu < %zu)",
                   ws_bytes, need);
    float* W = (float*)ws;
    {
        const dim3 grid(gx, (unsigned)sp.nsplit);
        GEOGCN_GEMM_LAUNCH(GEOGCN_ACT_NONE, 1, grid, W, ldw, sp.kchunk);
    }
    const dim3 rgrid((unsigned)cdiv(M * N, TPB));
#define GEOGCN_RED(ACT)                                                                                 \
    hipLaunchKernelGGL((splitk_reduce_kernel<ACT>), rgrid, dim3(TPB), 0, st, M, N, sp.nsplit, W, ldw, C, \
                       ldc, bias, accumulate)
    if (act == GEOGCN_ACT_TANH) GEOGCN_RED(GEOGCN_ACT_TANH);
    else if (act == GEOGCN_ACT_SIGMOID) GEOGCN_RED(GEOGCN_ACT_SIGMOID);
    else GEOGCN_RED(GEOGCN_ACT_NONE);
#undef GEOGCN_RED
    GEOGCN_LAUNCH_CHECK("splitk_reduce_kernel");
    return 0;
#undef GEOGCN_GEMM_LAUNCH
}

constexpr int kBM = 128, kBN = 128;

}  // namespace
}  // namespace geogcn

using namespace geogcn;

extern "C" {

size_t geogcn_gemm_workspace_bytes(int32_t transA, int32_t transB, int64_t M, int64_t N, int64_t K) {
    (void)transB;
    if (!transA || M <= 0 || N <= 0 || K <= 0) return 0;
    const SplitPlan sp = plan_split<kBM, kBN>(M, N, K);
    if (sp.nsplit <= 1) return 0;
    return (size_t)sp.nsplit * (size_t)M * (size_t)N * sizeof(float);
}

int geogcn_gemm_f32(int32_t transA, int32_t transB, int64_t M, int64_t N, int64_t K, const float* A,
                    int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, const float* bias,
                    int32_t act, int32_t accumulate, void* ws, size_t ws_bytes, void* stream) {
    GEOGCN_REQUIRE(M >= 0 && N >= 0 && K >= 0, GEOGCN_E_SIZE, "gemm_f32: negative size");
    if (M == 0 || N == 0) return 0;
    GEOGCN_REQUIRE(C && (K == 0 || (A && B)), GEOGCN_E_NULL, "gemm_f32: null pointer");
    GEOGCN_REQUIRE(act >= GEOGCN_ACT_NONE && act <= GEOGCN_ACT_SIGMOID, GEOGCN_E_ARG, "gemm_f32: unknown act %d", act);
    GEOGCN_REQUIRE(!(transA && transB), GEOGCN_E_ARG, "gemm_f32: transA && transB not supported");
    const int64_t a_cols = transA ? M : K, b_cols = transB ? K : N;
    GEOGCN_REQUIRE(lda >= a_cols && ldb >= b_cols && ldc >= N, GEOGCN_E_SIZE,
                   "gemm_f32: leading dimension too small (lda=%lld ldb=%lld ldc=%lld)", (long long)lda,
                   (long long)ldb, (long long)ldc);
    GEOGCN_REQUIRE(lda % 4 == 0 && ldb % 4 == 0 && aligned16(A) && aligned16(B), GEOGCN_E_ALIGN,
                   "gemm_f32: operands need 16-byte aligned bases and ld %% 4 == 0 (lda=%lld ldb=%lld)",
                   (long long)lda, (long long)ldb);
    hipStream_t st = (hipStream_t)stream;
    if (transA) return launch_gemm<kBM, kBN, true, false>(M, N, K, A, lda, B, ldb, C, ldc, bias, act, accumulate, ws, ws_bytes, st);
    if (transB) return launch_gemm<kBM, kBN, false, true>(M, N, K, A, lda, B, ldb, C, ldc, bias, act, accumulate, ws, ws_bytes, st);
    return launch_gemm<kBM, kBN, false, false>(M, N, K, A, lda, B, ldb, C, ldc, bias, act, accumulate, ws, ws_bytes, st);
}

}  // extern "C"
