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

__device__ __forceinline__ float4 load4_guard(const float* __restrict__ p, int64_t idx, int64_t lim4, bool row_ok) {
    // One predicated 16-byte load, no scalar tail: `lim4` is the limit of the contiguous dimension
    // rounded UP to a multiple of 4 (<= ld), so a float4 is either wholly inside or wholly outside.
    // Elements in [lim, lim4) are pad columns, which are zero by the geogcn.h convention.
    // (A branchy tail here made hipcc put s_waitcnt vmcnt(0) in front of every load of the tile.)
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row_ok && idx < lim4) r = *reinterpret_cast<const float4*>(p + idx);
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
        reg[i] = load4_guard(P + row * ld, k0 + f4 * 4, (Kend + 3) & ~(int64_t)3, row < Rtot);
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
        reg[i] = load4_guard(P + k * ld, c0 + c4 * 4, (Ctot + 3) & ~(int64_t)3, k < Kend);
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

struct GemmArgs {
    int64_t M, N, K;
    const float* A; int64_t lda;
    const float* B; int64_t ldb;
    float* C; int64_t ldc;
    const float* bias;
    int accumulate;
    int64_t kchunk;
    int n_mt, n_nt, n_split, xcd_order;
};

struct TileCoord {
    int64_t m0, n0, kbeg, kend;
    int nk, z;
    bool valid;
};

// Persistent blocks: block p walks its tile list p, p+G, ... (XCD-aware: the blocks of one XCD walk the
// N tiles / K slices of the same M tiles concurrently, so the A panel is shared through that XCD's L2).
__device__ __forceinline__ TileCoord decode_tile(const GemmArgs& a, int BM, int BN, int p, int G, int j) {
    TileCoord t;
    int mt, nt, z;
    if (a.xcd_order) {
        const int x = p % kNumXCD, q = p / kNumXCD, Q = G / kNumXCD;
        const int64_t u = (int64_t)q + (int64_t)j * Q;
        nt = (int)(u % a.n_nt);
        const int64_t rest = u / a.n_nt;
        z = (int)(rest % a.n_split);
        mt = (int)(rest / a.n_split) * kNumXCD + x;
    } else {
        const int64_t u = (int64_t)p + (int64_t)j * G;
        nt = (int)(u % a.n_nt);
        const int64_t rest = u / a.n_nt;
        mt = (int)(rest % a.n_mt);
        z = (int)(rest / a.n_mt);
        if (z >= a.n_split) mt = a.n_mt;           // past the end
    }
    t.valid = mt < a.n_mt;
    t.m0 = (int64_t)mt * BM;
    t.n0 = (int64_t)nt * BN;
    t.z = z;
    t.kbeg = (int64_t)z * a.kchunk;
    t.kend = min(a.K, t.kbeg + a.kchunk);
    t.nk = t.valid ? (int)((t.kend - t.kbeg + BK - 1) / BK) : 0;
    if (t.nk <= 0) t.valid = false;
    return t;
}

// MODE 0: C = act(acc + bias) [+ C if accumulate];  MODE 1: split-K slab z (raw partial sums)
// The k-loop is FLATTENED across the block's tiles: the global loads of stage s+1 are always in
// flight during the MFMAs of stage s, also across a tile boundary, so the short K = 300 contractions
// of the GCN (10 stages per tile) pay no per-tile prologue.
template <int BM, int BN, bool AT, bool BT, int ACT, int MODE>
__global__ __launch_bounds__(TPB, 2) void gemm_kernel(const GemmArgs a) {
    using Cfg = GemmCfg<BM, BN, AT, BT>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int p = blockIdx.x, G = gridDim.x;

    const int lane = threadIdx.x & 63;
    const int wid = threadIdx.x >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int li = lane & 15, lg = lane >> 4;

    f32x4 acc[Cfg::MR][Cfg::NR];
#pragma unroll
    for (int i = 0; i < Cfg::MR; ++i)
#pragma unroll
        for (int jn = 0; jn < Cfg::NR; ++jn) acc[i][jn] = f32x4{0.f, 0.f, 0.f, 0.f};

    // two register sets: while stage s is multiplied, stage s+1 waits in one set to be written to LDS and
    // the global loads of stage s+2 land in the other (two stages of HBM latency tolerance)
    float4 ra0[Cfg::kAIters], rb0[Cfg::kBIters], ra1[Cfg::kAIters], rb1[Cfg::kBIters];
    auto gload = [&](float4 (&ra)[Cfg::kAIters], float4 (&rb)[Cfg::kBIters], const TileCoord& t, int kt) {
        const int64_t k0 = t.kbeg + (int64_t)kt * BK;
        if constexpr (AT) gload_kstrided<BM>(ra, a.A, a.lda, t.m0, a.M, k0, t.kend);
        else gload_kcontig<BM>(ra, a.A, a.lda, t.m0, a.M, k0, t.kend);
        if constexpr (BT) gload_kcontig<BN>(rb, a.B, a.ldb, t.n0, a.N, k0, t.kend);
        else gload_kstrided<BN>(rb, a.B, a.ldb, t.n0, a.N, k0, t.kend);
    };
    auto sstore = [&](int buf, const float4 (&ra)[Cfg::kAIters], const float4 (&rb)[Cfg::kBIters]) {
        float* As = smem + buf * Cfg::kStageFloats;
        float* Bs = As + Cfg::kAFloats;
        if constexpr (AT) sstore_kstrided<BM>(As, ra);
        else sstore_kcontig<BM>(As, ra);
        if constexpr (BT) sstore_kcontig<BN>(Bs, rb);
        else sstore_kstrided<BN>(Bs, rb);
    };

    int cj = 0, ckt = 0;                     // compute cursor (tile index in my list, stage)
    TileCoord ct = decode_tile(a, BM, BN, p, G, 0);
    if (!ct.valid) return;
    int lj = 0, lkt = 0;                     // load cursor
    TileCoord lt = ct;
    auto advance_load = [&]() {
        if (++lkt == lt.nk) { lt = decode_tile(a, BM, BN, p, G, ++lj); lkt = 0; }
    };
    gload(ra0, rb0, lt, 0);
    sstore(0, ra0, rb0);
    advance_load();
    bool pending = lt.valid;                 // stage s+1 sits in a register set, not yet in LDS
    if (pending) { gload(ra1, rb1, lt, lkt); advance_load(); }
    __syncthreads();
    int cur = 0;
    bool running = true;
    // one pipeline step; (la, lb) = set to load stage s+2 into, (sa, sb) = set holding stage s+1
    auto step = [&](float4 (&la)[Cfg::kAIters], float4 (&lb)[Cfg::kBIters], const float4 (&sa)[Cfg::kAIters],
                    const float4 (&sb)[Cfg::kBIters]) {
        const bool have_load = lt.valid;
        if (have_load) gload(la, lb, lt, lkt);
        const float* As = smem + cur * Cfg::kStageFloats;
        const float* Bs = As + Cfg::kAFloats;
        const int64_t k_stage = ct.kbeg + (int64_t)ckt * BK;
#pragma unroll
        for (int kk = 0; kk < BK; kk += 16) {
            if (kk > 0 && k_stage + kk >= ct.kend) break;      // K tail: nothing but zero padding left
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
        if (ckt == ct.nk - 1) {
            // epilogue of this tile: C/D layout of the 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + r
            float* Cout = a.C;
            if constexpr (MODE == 1) Cout = a.C + (int64_t)ct.z * a.M * a.ldc;
            // all loads of the epilogue (bias, old C when accumulating) are issued as independent
            // batches before their first use -- no load/wait/store chains
            float bcol[Cfg::NR];
#pragma unroll
            for (int jn = 0; jn < Cfg::NR; ++jn) {
                const int64_t col = ct.n0 + wn * (BN / 2) + jn * 16 + li;
                bcol[jn] = 0.f;
                if constexpr (MODE == 0) {
                    if (a.bias && col < a.N) bcol[jn] = a.bias[col];
                }
            }
#pragma unroll
            for (int i = 0; i < Cfg::MR; ++i) {
                const int64_t row0 = ct.m0 + wm * (BM / 2) + i * 16 + lg * 4;
                float oldv[Cfg::NR][4];
                if constexpr (MODE == 0) {
                    if (a.accumulate) {
#pragma unroll
                        for (int jn = 0; jn < Cfg::NR; ++jn) {
                            const int64_t col = ct.n0 + wn * (BN / 2) + jn * 16 + li;
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                oldv[jn][r] = (row0 + r < a.M && col < a.N) ? Cout[(row0 + r) * a.ldc + col] : 0.f;
                        }
                    }
                }
#pragma unroll
                for (int jn = 0; jn < Cfg::NR; ++jn) {
                    const int64_t col = ct.n0 + wn * (BN / 2) + jn * 16 + li;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float x = acc[i][jn][r];
                        if constexpr (MODE == 0) {
                            x = apply_act<ACT>(x + bcol[jn]);
                            if (a.accumulate) x += oldv[jn][r];
                        }
                        if (row0 + r < a.M && col < a.N) Cout[(row0 + r) * a.ldc + col] = x;
                    }
                    acc[i][jn] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
        }
        if (pending) sstore(cur ^ 1, sa, sb);
        __syncthreads();
        cur ^= 1;
        if (++ckt == ct.nk) {
            ct = decode_tile(a, BM, BN, p, G, ++cj);
            ckt = 0;
            if (!ct.valid) running = false;
        }
        pending = have_load;
        if (have_load) advance_load();
    };
    while (running) {
        step(ra0, rb0, ra1, rb1);            // set 1 holds s+1, set 0 is free for s+2
        if (!running) break;
        step(ra1, rb1, ra0, rb0);
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
    int grid;
};

template <int BM, int BN, bool AT, bool BT>
constexpr int blocks_per_cu() { return (2 * GemmCfg<BM, BN, AT, BT>::kLdsBytes <= 160 * 1024) ? 2 : 1; }

// grid = resident persistent blocks; transA additionally slices K so that (tiles x slices) fills the grid
template <int BM, int BN, bool AT, bool BT>
SplitPlan plan_grid(int64_t M, int64_t N, int64_t K) {
    const int64_t tiles = cdiv(M, BM) * cdiv(N, BN);
    const int G = kNumCU * blocks_per_cu<BM, BN, AT, BT>();
    SplitPlan sp{1, cdiv(K, BK) * BK, 0};
    if (AT) {
        int64_t ns = std::max<int64_t>(1, G / tiles);
        const int64_t max_ns = std::max<int64_t>(1, K / (BK * 16));
        ns = std::min(ns, max_ns);
        sp.kchunk = cdiv(cdiv(K, ns), BK) * BK;
        sp.nsplit = (int)cdiv(K, sp.kchunk);
    }
    const int64_t total = tiles * sp.nsplit;
    sp.grid = (int)std::min<int64_t>(G, cdiv(total, kNumXCD) * kNumXCD);
    return sp;
}

template <int BM, int BN, bool AT, bool BT>
int launch_gemm(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B, int64_t ldb,
                float* C, int64_t ldc, const float* bias, int act, int accumulate, void* ws, size_t ws_bytes,
                hipStream_t st) {
    using Cfg = GemmCfg<BM, BN, AT, BT>;
    const SplitPlan sp = plan_grid<BM, BN, AT, BT>(M, N, K);
    GemmArgs a{M, N, K, A, lda, B, ldb, C, ldc, bias, accumulate, sp.kchunk, (int)cdiv(M, BM), (int)cdiv(N, BN),
               sp.nsplit, 0};
    a.xcd_order = (a.n_mt >= 4 * kNumXCD) ? 1 : 0;
    const dim3 grid((unsigned)sp.grid);
#define GEOGCN_GEMM_LAUNCH(ACT, MODE)                                                                    \
    do {                                                                                                  \
        auto kern = gemm_kernel<BM, BN, AT, BT, ACT, MODE>;                                               \
        static bool attr_done = false;                                                                    \
        if (!attr_done) {                                                                                 \
            GEOGCN_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                           (int)Cfg::kLdsBytes));                                         \
            attr_done = true;                                                                             \
        }                                                                                                 \
        hipLaunchKernelGGL(kern, grid, dim3(TPB), Cfg::kLdsBytes, st, a);                                 \
        GEOGCN_LAUNCH_CHECK("gemm_kernel");                                                               \
    } while (0)

    if (sp.nsplit == 1) {
        if (act == GEOGCN_ACT_TANH) GEOGCN_GEMM_LAUNCH(GEOGCN_ACT_TANH, 0);
        else if (act == GEOGCN_ACT_SIGMOID) GEOGCN_GEMM_LAUNCH(GEOGCN_ACT_SIGMOID, 0);
        else GEOGCN_GEMM_LAUNCH(GEOGCN_ACT_NONE, 0);
        return 0;
    }
    // split-K slabs into the workspace, then the ordered combine
    const int64_t ldw = N;
    const size_t need = (size_t)sp.nsplit * (size_t)M * (size_t)ldw * sizeof(float);
    GEOGCN_REQUIRE(ws && ws_bytes >= need, GEOGCN_E_ARG, "gemm_f32: split-K workspace too small (%zu < %zu)",
                   ws_bytes, need);
    float* W = (float*)ws;
    a.C = W;
    a.ldc = ldw;
    a.bias = nullptr;
    a.accumulate = 0;
    GEOGCN_GEMM_LAUNCH(GEOGCN_ACT_NONE, 1);
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

// ---- tile selection ------------------------------------------------------------------------------
// 128 or 160 per dimension, whichever wastes fewer MFMA columns on padding (300 -> 2x160, 256 -> 2x128,
// 600 -> 4x160).  The long dimension of NN / NT always uses BM = 128 (thousands of tiles).
inline int pick_tile(int64_t n) {
    const int64_t w128 = cdiv(n, 128) * 128, w160 = cdiv(n, 160) * 160;
    return (w160 < w128) ? 160 : 128;
}

template <bool AT, bool BT>
int dispatch_tiles(int bm, int bn, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B,
                   int64_t ldb, float* C, int64_t ldc, const float* bias, int act, int accumulate, void* ws,
                   size_t ws_bytes, hipStream_t st) {
#define GEOGCN_T(BM_, BN_)                                                                                   \
    if (bm == BM_ && bn == BN_)                                                                              \
        return launch_gemm<BM_, BN_, AT, BT>(M, N, K, A, lda, B, ldb, C, ldc, bias, act, accumulate, ws, ws_bytes, st);
    GEOGCN_T(128, 128)
    GEOGCN_T(128, 160)
    if constexpr (BT) {
        GEOGCN_T(96, 160)           // two k-contiguous images of 128+160 rows would not fit twice in 160 KB
    }
    if constexpr (AT) {
        GEOGCN_T(160, 128)
        GEOGCN_T(160, 160)
    }
#undef GEOGCN_T
    set_error("gemm_f32: no kernel for tile %dx%d", bm, bn);
    return GEOGCN_E_ARG;
}

template <int BM, int BN>
size_t splitk_ws_bytes(int64_t M, int64_t N, int64_t K) {
    const SplitPlan sp = plan_grid<BM, BN, true, false>(M, N, K);
    return sp.nsplit <= 1 ? 0 : (size_t)sp.nsplit * (size_t)M * (size_t)N * sizeof(float);
}

}  // namespace
}  // namespace geogcn

using namespace geogcn;

extern "C" {

size_t geogcn_gemm_workspace_bytes(int32_t transA, int32_t transB, int64_t M, int64_t N, int64_t K,
                                   int32_t precision) {
    (void)transB;
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    if (!transA) return precision == GEOGCN_GEMM_F32 ? 0 : gemm_bf16_workspace_bytes(precision, N, K);
    const int bm = pick_tile(M), bn = pick_tile(N);
    if (bm == 128 && bn == 128) return splitk_ws_bytes<128, 128>(M, N, K);
    if (bm == 128 && bn == 160) return splitk_ws_bytes<128, 160>(M, N, K);
    if (bm == 160 && bn == 128) return splitk_ws_bytes<160, 128>(M, N, K);
    return splitk_ws_bytes<160, 160>(M, N, K);
}

int geogcn_gemm_f32(int32_t transA, int32_t transB, int64_t M, int64_t N, int64_t K, const float* A,
                    int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, const float* bias,
                    int32_t act, int32_t accumulate, int32_t precision, void* ws, size_t ws_bytes, void* stream) {
    GEOGCN_REQUIRE(M >= 0 && N >= 0 && K >= 0, GEOGCN_E_SIZE, "gemm_f32: negative size");
    GEOGCN_REQUIRE(precision >= GEOGCN_GEMM_F32 && precision <= GEOGCN_GEMM_BF16, GEOGCN_E_ARG,
                   "gemm_f32: unknown precision %d", precision);
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
    if (!transA && precision != GEOGCN_GEMM_F32 && K > 0)
        return gemm_bf16_dispatch(precision, transB, M, N, K, A, lda, B, ldb, C, ldc, bias, act, accumulate, ws, ws_bytes, st);
    const int bn = pick_tile(N);
    if (transA)
        return dispatch_tiles<true, false>(pick_tile(M), bn, M, N, K, A, lda, B, ldb, C, ldc, bias, act, accumulate, ws,
                                           ws_bytes, st);
    if (transB)
        return dispatch_tiles<false, true>(bn == 160 ? 96 : 128, bn, M, N, K, A, lda, B, ldb, C, ldc, bias, act, accumulate,
                                           ws, ws_bytes, st);
    return dispatch_tiles<false, false>(128, bn, M, N, K, A, lda, B, ldb, C, ldc, bias, act, accumulate, ws, ws_bytes, st);
}

}  // extern "C"
