// bf16-MFMA GEMM for the activations x weights contractions of the GCN hot path (gfx950):
//   C[M x N] = act(A[M x K] . B + bias) [+ C]     A fp32 row-major (the N_nodes x hid activations),
//   B = W (K x N, forward, reference gcnmodel.py:126,149,285) or W given as N x K (dH = dZ . W^T).
// fp32 in HBM on both sides; the operands are converted on their way into LDS and multiplied with
// v_mfma_f32_16x16x32_bf16 (fp32 accumulate), 16x the rate of the fp32-input MFMA, in one of two modes:
//   NS = 1  "bf16"    one bf16 term per operand (BASELINE config 5: bf16 H.W, fp32 accumulate);
//   NS = 3  "bf16x3"  every fp32 value is split EXACTLY into three bf16 terms (8+8+8 mantissa bits:
//           a = a1 + a2 + a3 to 2^-24) and the product is formed from the six largest cross terms
//           a1b1 + a1b2 + a2b1 + a2b2 + a1b3 + a3b1 (each bf16 x bf16 product is exact in fp32; the
//           dropped terms are O(2^-24 |a||b|)), i.e. fp32-class accuracy at 6/16 of the fp32-MFMA cost.
//           With it the contraction stops being MFMA-bound and runs at the HBM streaming rate of A and C.
// The weights are tiny (<= 600 x 600): a prep kernel writes their bf16 planes [NS][N][Kp] (k-contiguous,
// zero padded to Kp = roundup32(K)) once per call, so the main kernel never transposes in LDS.
// Same persistent, flattened-k-pipeline structure as gemm.hip (one LDS image, two blocks per CU).
#include "common.h"

#include <stdlib.h>

#include <algorithm>

namespace geogcn {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

constexpr int TPB = 256;
constexpr int BKH = 32;                  // k per stage = one MFMA depth
constexpr int ROWB = 80;                 // bytes per LDS row: 32 bf16 (64 B) + 16 B pad

__device__ __forceinline__ uint32_t f2u(float x) { return __float_as_uint(x); }
__device__ __forceinline__ float u2f(uint32_t u) { return __uint_as_float(u); }

// round-to-nearest-even fp32 -> bf16: gfx950's v_cvt_pk_bf16_f32 converts two values per instruction (the integer
// sequence (u + 0x7fff + lsb) >> 16 it replaces was five VALU instructions per value; same result for finite inputs)
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t bf16_pack(float lo, float hi) {
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ uint32_t bf16_rne(float x) { return bf16_pack(x, 0.f) & 0xffffu; }

// x -> NS bf16 terms; the residuals are exact in fp32 (Dekker-style splitting)
template <int NS>
__device__ __forceinline__ void split_bf16(float x, uint32_t (&t)[NS]) {
    t[0] = bf16_rne(x);
    if constexpr (NS > 1) {
        const float r1 = x - u2f(t[0] << 16);
        t[1] = bf16_rne(r1);
        if constexpr (NS > 2) {
            const float r2 = r1 - u2f(t[1] << 16);
            t[2] = bf16_rne(r2);
        }
    }
}

// ---- weights -> bf16 planes [NS][N][Kp], k-contiguous -------------------------------------------------
template <int NS>
__global__ __launch_bounds__(TPB) void prep_b_planes_kernel(const float* __restrict__ W, int64_t ldw, int K, int N,
                                                            int Kp, int b_is_nk, unsigned short* __restrict__ out) {
    const int64_t total = (int64_t)N * Kp;
    for (int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x; e < total; e += (int64_t)gridDim.x * TPB) {
        const int n = (int)(e / Kp), k = (int)(e - (int64_t)n * Kp);
        float x = 0.f;
        if (k < K) x = b_is_nk ? W[(int64_t)n * ldw + k] : W[(int64_t)k * ldw + n];
        uint32_t t[NS];
        split_bf16<NS>(x, t);
#pragma unroll
        for (int p = 0; p < NS; ++p) out[(int64_t)p * total + e] = (unsigned short)t[p];
    }
}

struct Bf16Args {
    int64_t M, N, K;
    const float* A; int64_t lda;
    const unsigned short* Bp;           // planes [NS][N][Kp]
    int Kp;
    void* C; int64_t ldc;               // fp32, or bf16 bit patterns when c_bf16
    const float* bias;
    int accumulate;
    int n_mt, n_nt;
    int c_bf16;
    int64_t n_store;                    // columns written per row (pads beyond N as zeros)
    int panel_w;                        // > 0: C is laid out as feature panels [N / panel_w][panel_R][panel_w] (gemm.hip)
    int64_t panel_R;
    const float* gateG = nullptr; int64_t ldg = 0; const float* gateT = nullptr; int64_t ldt = 0;      // whole-rows kernel only (common.h GateOps)
    // whole-rows kernel only: a SECOND column segment of the same launch (the highway block's H . [Wh | Wt]: segment 0 = the
    // fields above, without activation; segment 1 = these, with the launch's activation) -- A is read and rounded once
    int n_nseg = 1;
    int passes0 = 0, passes1 = 0;
    const unsigned short* Bp1 = nullptr; void* C1 = nullptr; int64_t ldc1 = 0; const float* bias1 = nullptr; int64_t N1 = 0;
    int c_bf16_1 = 0; int64_t n_store1 = 0;
    // whole-rows kernel, KCAT instances only (round 6): a SECOND reduction segment -- C = A . op(B) + A1 . op(B1), one accumulator
    // (dH = dZ . Wh^T + dU . Wt^T of the highway block): A1's tile sits behind A's in the LDS rows, B1's fragments behind B's in k
    const float* A1 = nullptr; int64_t lda1 = 0; int64_t K1 = 0;
};

template <int BM, int BN, int NS, int NT>
struct BCfg {
    static constexpr int kWavesM = NT / 128;                 // waves along M (2 or 4); always 2 along N
    static constexpr int kARows = BM, kBRows = BN;
    static constexpr int kABytes = NS * BM * ROWB;
    static constexpr int kBBytes = NS * BN * ROWB;
    static constexpr int kStageBytes = kABytes + kBBytes;
    // double buffered (one barrier per stage) when two blocks per CU still fit, else one image + two barriers
    // (the 8-wave x3 variant is limited to one block per CU by registers: it double-buffers whenever 2 images fit)
    static constexpr bool kDouble = (NT == 512) ? (2 * kStageBytes <= 160 * 1024) : (4 * kStageBytes <= 160 * 1024);
    static constexpr int kLdsBytes = (kDouble ? 2 : 1) * kStageBytes;
    static constexpr int MR = BM / (16 * kWavesM), NR = BN / 32;
    static constexpr int kAIters = BM / (NT / 8);           // float4 per thread per stage (8 f4 per row)
    static constexpr int kBVec = NS * BN * 4;               // uint4 per stage (4 per row per plane)
    static constexpr int kBIters = (kBVec + NT - 1) / NT;
};

constexpr uint32_t kOob = 0x80000000u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tn_rsrc(const float* base, int64_t bytes) {
    const uint32_t n = bytes <= 0 ? 0u : (bytes > 0x7FFFFFFFll ? 0x7FFFFFFFu : (uint32_t)bytes);
    const uint64_t b = reinterpret_cast<uint64_t>(base);
    const uint64_t bu = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(b >> 32)) << 32) |
                        (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)b);
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(bu), 0, __builtin_amdgcn_readfirstlane(n), 0x00020000);
}
__device__ __forceinline__ float4 tn_load4(__amdgpu_buffer_rsrc_t r, uint32_t off) {
    const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0));
    return make_float4(v.x, v.y, v.z, v.w);
}


// PROBE (ablation, GEOGCN_BF16_PROBE; 0 in normal use): 1 = no global loads in the loop, 2 = no LDS stores,
// 4 = no fragment reads / MFMAs, 8 = no barrier
template <int BM, int BN, int NS, int ACT, int NT, int PROBE = 0>
__global__ __launch_bounds__(NT, 2) void gemm_bf16_kernel(const Bf16Args a) {
    using Cfg = BCfg<BM, BN, NS, NT>;
    constexpr int kRowsPerPass = NT / 8;
    constexpr int kWaveRows = BM / Cfg::kWavesM;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int p = blockIdx.x, G = gridDim.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int li = lane & 15, lg = lane >> 4;
    const int nk = a.Kp / BKH;

    f32x4 acc[Cfg::MR][Cfg::NR];
#pragma unroll
    for (int i = 0; i < Cfg::MR; ++i)
#pragma unroll
        for (int j = 0; j < Cfg::NR; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // XCD-aware persistent tile walk (as gemm.hip): the blocks of one XCD share the A panel through L2.
    // Tile j of this block is (mt, nt) = decode(j); valid for j < n_my (mt grows with j).
    const int x = p % kNumXCD, q = p / kNumXCD, Q = G / kNumXCD;
    auto decode = [&](int j, int& mt, int& nt) {
        const int64_t u = (int64_t)q + (int64_t)j * Q;
        nt = __builtin_amdgcn_readfirstlane((int)(u % a.n_nt));
        mt = __builtin_amdgcn_readfirstlane((int)(u / a.n_nt) * kNumXCD + x);
    };
    int n_my = 0;
    {
        int mt, nt;
        for (;; ++n_my) {
            decode(n_my, mt, nt);
            if (mt >= a.n_mt) break;
        }
    }
    if (n_my == 0) return;

    // A (HBM stream) is prefetched TWO stages ahead in two register sets, B (L2-resident weight planes) one
    // stage ahead.  Everything arrives through raw buffer loads with tile-rebased descriptors (rows past the end
    // and stages past this block's list read as zeros): the loop body has no data-dependent branches, so the
    // wait counts are exact (the branchy version drained the memory pipeline at every join, see gemm.hip).
    float4 ra0[Cfg::kAIters], ra1[Cfg::kAIters];
    uint4 rb[Cfg::kBIters];
    const int a_f4 = tid & 7, a_rr = tid >> 3;
    auto gloadA = [&](float4 (&ra)[Cfg::kAIters], int j, int kt) {
        int mt, nt;
        decode(j, mt, nt);
        const bool valid = j < n_my;
        const int64_t m0 = (int64_t)mt * BM;
        const int64_t k0 = (int64_t)kt * BKH;
        const int64_t rows = valid ? a.M - m0 : 0;
        const __amdgpu_buffer_rsrc_t rs = tn_rsrc(a.A + m0 * a.lda + k0, (rows * a.lda - k0) * 4);
        const bool k_ok = k0 + a_f4 * 4 < ((a.K + 3) & ~(int64_t)3);      // pad columns of A are zero (geogcn.h)
        const uint32_t ld4 = (uint32_t)a.lda * 4u;
#pragma unroll
        for (int i = 0; i < Cfg::kAIters; ++i) {
            const uint32_t off = (uint32_t)(a_rr + kRowsPerPass * i) * ld4 + (uint32_t)a_f4 * 16u;
            ra[i] = tn_load4(rs, k_ok ? off : kOob);
        }
    };
    auto gloadB = [&](int j, int kt) {
        int mt, nt;
        decode(j, mt, nt);
        const bool valid = j < n_my;
        const int64_t n0 = (int64_t)nt * BN;
        // planes [NS][N][Kp] bf16: one descriptor over all of them (<= a few MB)
        const __amdgpu_buffer_rsrc_t rs = tn_rsrc(reinterpret_cast<const float*>(a.Bp),
                                                  valid ? (int64_t)NS * a.N * a.Kp * 2 : 0);
#pragma unroll
        for (int i = 0; i < Cfg::kBIters; ++i) {
            const int e = tid + NT * i;
            const int pl = e / (BN * 4), r = (e / 4) % BN, c = e & 3;
            const int64_t n = n0 + r;
            const bool ok = e < Cfg::kBVec && n < a.N;
            const uint32_t off = (uint32_t)((((int64_t)pl * a.N + n) * a.Kp + (int64_t)kt * BKH + c * 8) * 2);
            const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(ok ? off : kOob), 0, 0));
            rb[i] = __builtin_bit_cast(uint4, v);
        }
    };
    auto sstore = [&](int buf, const float4 (&ra)[Cfg::kAIters]) {
        unsigned char* As = smem_raw + (Cfg::kDouble ? buf : 0) * Cfg::kStageBytes;
        unsigned char* Bs = As + Cfg::kABytes;
#pragma unroll
        for (int i = 0; i < Cfg::kAIters; ++i) {
            const int row = a_rr + kRowsPerPass * i;
            if constexpr (NS == 1) {
                uint2 w;
                w.x = bf16_pack(ra[i].x, ra[i].y);
                w.y = bf16_pack(ra[i].z, ra[i].w);
                *reinterpret_cast<uint2*>(As + row * ROWB + a_f4 * 8) = w;
            } else {
                const float xs[4] = {ra[i].x, ra[i].y, ra[i].z, ra[i].w};
                uint32_t t[4][NS];
#pragma unroll
                for (int e = 0; e < 4; ++e) split_bf16<NS>(xs[e], t[e]);
#pragma unroll
                for (int pl = 0; pl < NS; ++pl) {
                    uint2 w;
                    w.x = t[0][pl] | (t[1][pl] << 16);
                    w.y = t[2][pl] | (t[3][pl] << 16);
                    *reinterpret_cast<uint2*>(As + (pl * BM + row) * ROWB + a_f4 * 8) = w;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < Cfg::kBIters; ++i) {
            const int e = tid + NT * i;
            if (e < Cfg::kBVec) {
                const int pl = e / (BN * 4), r = (e / 4) % BN, c = e & 3;
                *reinterpret_cast<uint4*>(Bs + (pl * BN + r) * ROWB + c * 16) = rb[i];
            }
        }
    };

    // cursors: c = stage being multiplied, l = next A stage to load (two ahead), pj/pkt = stage c+1 (its B)
    int cj = 0, ckt = 0, lj = 0, lkt = 0, pj = 0, pkt = 0;
    auto adv = [&](int& j, int& kt) {
        if (++kt == nk) { kt = 0; ++j; }
    };
    gloadA(ra0, 0, 0);
    gloadB(0, 0);
    sstore(0, ra0);
    adv(lj, lkt);
    pj = lj; pkt = lkt;                       // stage 1
    gloadA(ra1, lj, lkt);
    adv(lj, lkt);                             // stage 2
    __syncthreads();
    int cur = 0;
    int cm_t, cn_t;
    decode(0, cm_t, cn_t);
    int64_t cm0 = (int64_t)cm_t * BM, cn0 = (int64_t)cn_t * BN;
    // (xa) = free A set, receives stage s+2;  (ya) = A set holding stage s+1
    auto step = [&](float4 (&xa)[Cfg::kAIters], const float4 (&ya)[Cfg::kAIters]) {
        if (!(PROBE & 1)) gloadA(xa, lj, lkt);
        adv(lj, lkt);
        if (!(PROBE & 1)) gloadB(pj, pkt);
        const unsigned char* As = smem_raw + (Cfg::kDouble ? cur : 0) * Cfg::kStageBytes;
        const unsigned char* Bs = As + Cfg::kABytes;
        // ---- MFMAs on the resident stage ----
        // the A fragments of the stage are read ONCE and kept (MR x NS x 4 VGPRs), the B fragment of a column tile is read
        // once per column tile: MR + NR ds_read_b128 per stage and plane instead of NR x (1 + MR).  (Until round 3 the A
        // fragments were re-read for every column tile to save registers: at one 16x16x32 bf16 MFMA = 16 cycles the 25
        // reads per 20 MFMAs of the 128 x 160 tile made the LDS port, not the matrix pipe, the busiest unit.)
        bf16x8 af[Cfg::MR][NS];
#pragma unroll
        for (int i = 0; i < ((PROBE & 4) ? 0 : Cfg::MR); ++i)
#pragma unroll
            for (int pl = 0; pl < NS; ++pl)
                af[i][pl] = *reinterpret_cast<const bf16x8*>(As + (pl * BM + wm * kWaveRows + i * 16 + li) * ROWB + lg * 16);
#pragma unroll
        for (int j = 0; j < ((PROBE & 4) ? 0 : Cfg::NR); ++j) {
            bf16x8 bf[NS];
#pragma unroll
            for (int pl = 0; pl < NS; ++pl)
                bf[pl] = *reinterpret_cast<const bf16x8*>(Bs + (pl * BN + wn * (BN / 2) + j * 16 + li) * ROWB + lg * 16);
#pragma unroll
            for (int i = 0; i < Cfg::MR; ++i) {
                f32x4 c = acc[i][j];
                // operands swapped (B fragment first): the 16x16 product comes out transposed, so a lane owns 4
                // CONSECUTIVE COLUMNS of one row of C and the epilogue stores 16 (fp32) / 8 (bf16) bytes at once
                if constexpr (NS == 3) {        // smallest terms first
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[0], af[i][2], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[2], af[i][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[1], af[i][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[0], af[i][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[1], af[i][0], c, 0, 0, 0);
                }
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[0], af[i][0], c, 0, 0, 0);
                acc[i][j] = c;
            }
        }
        if (ckt == nk - 1) {
            // lane (li, lg) holds C[row = li][col = 4*lg + r], r = 0..3, of each 16x16 sub-tile.  Columns in
            // [N, n_store) are pad columns and are written as zeros (n_store = roundup4(N), or roundup8(N) when C
            // is stored as bf16 for the SpMM, which reads 8 features at a time).
            float bcol[Cfg::NR][4];
#pragma unroll
            for (int j = 0; j < Cfg::NR; ++j) {
                const int64_t col0 = cn0 + wn * (BN / 2) + j * 16 + lg * 4;
#pragma unroll
                for (int r = 0; r < 4; ++r) bcol[j][r] = (a.bias && col0 + r < a.N) ? a.bias[col0 + r] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < Cfg::MR; ++i) {
                const int64_t row = cm0 + wm * kWaveRows + i * 16 + li;
                const bool row_ok = row < a.M;
                float4 oldv[Cfg::NR];
                if (a.accumulate) {          // (fp32 C only)
#pragma unroll
                    for (int j = 0; j < Cfg::NR; ++j) {
                        const int64_t col0 = cn0 + wn * (BN / 2) + j * 16 + lg * 4;
                        oldv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (row_ok && col0 < a.N) oldv[j] = *reinterpret_cast<const float4*>((const float*)a.C + row * a.ldc + col0);
                    }
                }
#pragma unroll
                for (int j = 0; j < Cfg::NR; ++j) {
                    const int64_t col0 = cn0 + wn * (BN / 2) + j * 16 + lg * 4;
                    float x[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) x[r] = apply_act<ACT>(acc[i][j][r] + bcol[j][r]);
                    if (a.accumulate) { x[0] += oldv[j].x; x[1] += oldv[j].y; x[2] += oldv[j].z; x[3] += oldv[j].w; }
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (col0 + r >= a.N) x[r] = 0.f;
                    if (row_ok && col0 < a.n_store) {
                        int64_t off = row * a.ldc + col0;
                        if (a.panel_w) {
                            const int64_t q = col0 / a.panel_w;
                            off = (q * a.panel_R + row) * a.panel_w + (col0 - q * a.panel_w);
                        }
                        if (a.c_bf16) {
                            uint2 w;
                            w.x = bf16_pack(x[0], x[1]);
                            w.y = bf16_pack(x[2], x[3]);
                            *reinterpret_cast<uint2*>((unsigned short*)a.C + off) = w;
                        } else {
                            *reinterpret_cast<float4*>((float*)a.C + off) = make_float4(x[0], x[1], x[2], x[3]);
                        }
                    }
                    acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
        }
        if constexpr (!Cfg::kDouble) __syncthreads();     // single image: everybody done reading first
        if (!(PROBE & 2)) sstore(cur ^ 1, ya);
        if (!(PROBE & 8)) __syncthreads();
        cur ^= 1;
        adv(pj, pkt);
        if (++ckt == nk) {
            ckt = 0;
            ++cj;
            decode(cj, cm_t, cn_t);
            cm0 = (int64_t)cm_t * BM;
            cn0 = (int64_t)cn_t * BN;
        }
    };
    while (true) {
        step(ra0, ra1);
        if (cj >= n_my) break;
        step(ra1, ra0);
        if (cj >= n_my) break;
    }
}

template <int BM, int BN, int NS, int NT>
int launch_bf16(const Bf16Args& a, int act, hipStream_t st) {
    using Cfg = BCfg<BM, BN, NS, NT>;
    const int per_cu = (2 * Cfg::kLdsBytes <= 160 * 1024) ? 2 : 1;
    const int64_t tiles = (int64_t)a.n_mt * a.n_nt;
    const int G = (int)std::min<int64_t>((int64_t)kNumCU * per_cu, cdiv(tiles, kNumXCD) * kNumXCD);
#define GEOGCN_L(ACT)                                                                                          \
    do {                                                                                                        \
        auto kern = gemm_bf16_kernel<BM, BN, NS, ACT, NT>;                                                          \
        static LdsAttrOnce lds_once;                                                                          \
        if (const int rc_ = lds_once.ensure((const void*)kern, (int)(Cfg::kLdsBytes))) return rc_; \
        hipLaunchKernelGGL(kern, dim3((unsigned)G), dim3(NT), Cfg::kLdsBytes, st, a);                          \
        GEOGCN_LAUNCH_CHECK("gemm_bf16_kernel");                                                                \
    } while (0)
    if (act == GEOGCN_ACT_TANH) GEOGCN_L(GEOGCN_ACT_TANH);
    else if (act == GEOGCN_ACT_SIGMOID) GEOGCN_L(GEOGCN_ACT_SIGMOID);
    else GEOGCN_L(GEOGCN_ACT_NONE);
#undef GEOGCN_L
    return 0;
}


// ---- the skinny contractions of the bf16 configuration (K <= 608, N <= 640): whole rows of A per block ----------------
// gemm_bf16_kernel above stages 128 x 32 slices of A per barrier -- one 128-byte line per row and stage, 26 KB in flight per
// block -- and runs latency-bound at 15 % of the bf16 pipe (profiles/r02_c_bf16_gemm.md).  For the shapes the GCN actually has
// (N_nodes x 600 x 600, x 300 x 300, x 600 x 256 ...) this kernel takes 64 WHOLE rows of A per block instead: one contiguous read
// (154 KB at K = 600, every load issued before the first is waited for), rounded to bf16 into LDS once (<= 79 KB: two blocks per
// CU), then every wave multiplies all 64 rows by its own columns -- 5 column tiles per pass, up to two passes -- with B fragments
// read straight from the L2-resident weights, which the prep kernel lays out in FRAGMENT order ([column tile][k-step][lane][8]:
// a wave's fragment load is 1 KB of consecutive bytes; with row-major planes it touched 16 half lines: 0.82 against 0.66 ms in
// tools/micro/bf16_astat.hip).  No barrier inside a tile, A read from HBM exactly once.  Same MFMA, same k order, same epilogue
// arithmetic as gemm_bf16_kernel: the results are bit-identical.
template <int NS_>
__global__ __launch_bounds__(TPB) void prep_b_frag_kernel(const float* __restrict__ W, int64_t ldw, int K, int N, int NKs, int n_tiles,
                                                          int b_is_nk, unsigned short* __restrict__ out, int kstep_base = 0, int nk_total = 0) {
    if (nk_total == 0) nk_total = NKs;          // (a k-concatenated launch: this weight's k-steps start at kstep_base of nk_total per column tile)
    const int64_t total = (int64_t)n_tiles * NKs * 512;
    for (int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x; e < total; e += (int64_t)gridDim.x * TPB) {
        const int el = (int)(e & 7), lane = (int)((e >> 3) & 63);
        const int64_t f = e >> 9;
        const int kt = (int)(f % NKs), nt = (int)(f / NKs);
        const int n = nt * 16 + (lane & 15), k = kt * 32 + (lane >> 4) * 8 + el;
        float x = 0.f;
        if (k < K && n < N) x = b_is_nk ? W[(int64_t)n * ldw + k] : W[(int64_t)k * ldw + n];
        out[(((int64_t)nt * nk_total + kstep_base + kt) << 9) + (e & 511)] = (unsigned short)bf16_rne(x);
    }
}

constexpr int kRowsBM = 64, kRowsWCT = 5, kRowsDepth = 2;

// KCAT (k-concatenated launch, round 6): KP counts BOTH reduction segments (2 x 608 / 320 / 256) and the k loop walks the weights'
// fragments of both into the same accumulators.  KCAT = 1: the LDS rows hold [A row | A1 row] (157 KB at 2 x 608: one block per CU,
// each tile loaded once); KCAT = 2: the LDS rows hold ONE segment (79 KB: two blocks per CU), every column pass loads segment 0's tile,
// multiplies, then segment 1's -- the accumulators live across the reload, as in x3_rows_kernel's K chunks.
template <int KP, int ACT, bool GATE = false, int KCAT = 0>
__global__ __launch_bounds__(TPB, ((KCAT == 2 ? KP / 2 : KP) * 2 + 16) * kRowsBM > 80 * 1024 ? 1 : 2) void gemm_bf16_rows_kernel(const Bf16Args a, const int passes) {
    constexpr int BM = kRowsBM, WCT = kRowsWCT, DEPTH = kRowsDepth, D1 = DEPTH + 1;
    constexpr int NSEG = KCAT ? 2 : 1, KPS = KP / NSEG;      // reduction segments; padded depth of one
    constexpr int KL = KCAT == 2 ? KPS : KP;    // k columns an LDS row holds
    constexpr int PITCH = KL * 2 + 16;          // bytes per LDS row: an odd multiple of 16 -> conflict-free ds_read_b128
    constexpr int F4R = KPS / 4;                // float4 per row and segment
    constexpr int ITERS = BM * F4R / TPB;
    constexpr int CH = 2, IPC = ITERS / CH;     // two batches of loads: half the staging registers
    constexpr int MR = BM / 16, NK = KP / 32, NKS = KPS / 32;
    static_assert(BM * F4R % (TPB * CH) == 0 && (PITCH / 16) % 2 == 1, "a tile must divide over the block in two batches");
    extern __shared__ __attribute__((aligned(16))) unsigned char As[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int P = a.n_nseg == 2 ? a.passes0 + a.passes1 : passes;
    for (int mt = blockIdx.x; mt < a.n_mt; mt += gridDim.x) {
        const int64_t m0 = (int64_t)mt * BM;
        const int64_t rows = std::min<int64_t>(BM, a.M - m0);
        // segment sg's 64 x KPS tile: fp32 from HBM (one descriptor per tile: rows past M read as zeros in hardware), rounded to bf16 into
        // the LDS rows at byte `at`
        auto load_seg = [&](const int sg, const int at) {
            // (the offsets of the loads and LDS stores are the same for every tile: left alone, hipcc computes them once and keeps
            //  ~76 registers alive across the MFMA loop -- an opaque copy of the thread index makes them per-tile work)
            int tt = tid;
            asm volatile("" : "+v"(tt));
            const float* Ap = (KCAT && sg) ? a.A1 : a.A;
            const int64_t lda = (KCAT && sg) ? a.lda1 : a.lda;
            // pad columns of A up to roundup4(K) are zero (geogcn.h); beyond: not read
            const int K4 = (int)((((KCAT && sg) ? a.K1 : a.K) + 3) & ~(int64_t)3);
            const uint32_t ld4 = (uint32_t)lda * 4u;
            const __amdgpu_buffer_rsrc_t rs = tn_rsrc(Ap + m0 * lda, rows * lda * 4);
#pragma unroll
            for (int ch = 0; ch < CH; ++ch) {
                float4 v[IPC];
#pragma unroll
                for (int i = 0; i < IPC; ++i) {
                    const int idx = tt + TPB * (ch * IPC + i);
                    const int r = idx / F4R, c = idx - r * F4R;
                    v[i] = tn_load4(rs, c * 4 < K4 ? (uint32_t)r * ld4 + (uint32_t)c * 16u : kOob);
                }
#pragma unroll
                for (int i = 0; i < IPC; ++i) {
                    const int idx = tt + TPB * (ch * IPC + i);
                    const int r = idx / F4R, c = idx - r * F4R;
                    uint2 w;
                    w.x = bf16_pack(v[i].x, v[i].y);
                    w.y = bf16_pack(v[i].z, v[i].w);
                    *reinterpret_cast<uint2*>(As + r * PITCH + at + c * 8) = w;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        if constexpr (KCAT != 2) {
            load_seg(0, 0);
            if constexpr (KCAT == 1) load_seg(1, KPS * 2);
            __syncthreads();
        }
#pragma unroll 1
        for (int ps = 0; ps < P; ++ps) {
            // (segment, pass inside it: wave-uniform)
            const int seg = (a.n_nseg == 2 && ps >= a.passes0) ? 1 : 0;
            const int lps = seg ? ps - a.passes0 : ps;
            const int spasses = a.n_nseg == 2 ? (seg ? a.passes1 : a.passes0) : passes;
            const int tile0 = (wid * spasses + lps) * WCT;        // first of this wave's column tiles in this pass
            const int64_t ncol0 = (int64_t)tile0 * 16;
            const __amdgpu_buffer_rsrc_t brs = tn_rsrc(reinterpret_cast<const float*>(seg ? a.Bp1 : a.Bp), (int64_t)4 * spasses * WCT * NK * 1024);
            const int64_t Nseg = seg ? a.N1 : a.N, nstore = seg ? a.n_store1 : a.n_store, ldc = seg ? a.ldc1 : a.ldc;
            void* const Cseg = seg ? a.C1 : a.C;
            const float* const bias = seg ? a.bias1 : a.bias;
            const bool cb16 = (seg ? a.c_bf16_1 : a.c_bf16) != 0;
            const bool act_on = a.n_nseg == 1 || seg == 1;
            f32x4 acc[MR][WCT];
#pragma unroll
            for (int i = 0; i < MR; ++i)
#pragma unroll
                for (int j = 0; j < WCT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            // B fragments DEPTH k-steps ahead in a ring of DEPTH + 1 register sets.  The k loop stays ROLLED (DEPTH + 1 steps per
            // trip; ring slots are compile-time constants inside a trip) and the scheduler is fenced per step, so the requests stay
            // one set per step; requests past the last step re-read the last one (no branch around a load)
            auto bload = [&](bf16x8 (&b)[WCT], int kt, int lim) {
                const int kk = kt < lim ? kt : lim - 1;
#pragma unroll
                for (int j = 0; j < WCT; ++j)
                    b[j] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(brs, lane * 16, ((tile0 + j) * NK + kk) * 1024, 0));
            };
            auto kstep = [&](const bf16x8 (&b)[WCT], int kt, int kl) {          // kl: the k-step inside the tile LDS holds
                bf16x8 af[MR];
#pragma unroll
                for (int i = 0; i < MR; ++i) af[i] = *reinterpret_cast<const bf16x8*>(As + (i * 16 + li) * PITCH + kl * 64 + lg * 16);
#pragma unroll
                for (int j = 0; j < WCT; ++j)
#pragma unroll
                    for (int i = 0; i < MR; ++i)
                        // operands swapped (B fragment first): the 16 x 16 product comes out transposed, a lane owns 4 consecutive
                        // columns of one row of C
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], af[i], acc[i][j], 0, 0, 0);
            };
            if constexpr (KCAT == 2) {
                // one reduction segment at a time: its tile into LDS (the ring of B fragments is not alive across the reload: the staging
                // registers of the tile load take its place), then its k-steps into the same accumulators
#pragma unroll 1
                for (int sg = 0; sg < 2; ++sg) {
                    __syncthreads();          // everybody done with the tile before (the previous segment's, pass's or row tile's)
                    load_seg(sg, 0);
                    __syncthreads();
                    const int kb = sg * NKS;
                    bf16x8 ring[D1][WCT];
#pragma unroll
                    for (int d = 0; d < DEPTH; ++d) bload(ring[d], kb + d, kb + NKS);
#pragma unroll 1
                    for (int k0 = 0; k0 < NKS; k0 += D1) {
#pragma unroll
                        for (int u = 0; u < D1; ++u) {
                            bload(ring[(u + DEPTH) % D1], kb + k0 + u + DEPTH, kb + NKS);
                            if (k0 + u < NKS) kstep(ring[u], kb + k0 + u, k0 + u);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
            } else {
                bf16x8 ring[D1][WCT];
#pragma unroll
                for (int d = 0; d < DEPTH; ++d) bload(ring[d], d, NK);
#pragma unroll 1
                for (int k0 = 0; k0 < NK; k0 += D1) {
#pragma unroll
                    for (int u = 0; u < D1; ++u) {
                        bload(ring[(u + DEPTH) % D1], k0 + u + DEPTH, NK);
                        if (k0 + u < NK) kstep(ring[u], k0 + u, k0 + u);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            // epilogue: the arithmetic of gemm_bf16_kernel, in its order
            float bcol[WCT][4];
#pragma unroll
            for (int j = 0; j < WCT; ++j) {
                const int64_t col0 = ncol0 + j * 16 + lg * 4;
#pragma unroll
                for (int r = 0; r < 4; ++r) bcol[j][r] = (bias && col0 + r < Nseg) ? bias[col0 + r] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < MR; ++i) {
                const int64_t row = m0 + i * 16 + li;
                const bool row_ok = row < a.M;
                float4 oldv[WCT];
                if (a.accumulate) {          // (fp32 C only)
#pragma unroll
                    for (int j = 0; j < WCT; ++j) {
                        const int64_t col0 = ncol0 + j * 16 + lg * 4;
                        oldv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (row_ok && col0 < Nseg) oldv[j] = *reinterpret_cast<const float4*>((const float*)Cseg + row * ldc + col0);
                    }
                }
                float4 gv[GATE ? WCT : 1], tv[GATE ? WCT : 1];
                if constexpr (GATE) {        // C += G * (1 - T): the highway block's carry gradient (fp32 C)
#pragma unroll
                    for (int j = 0; j < WCT; ++j) {
                        const int64_t col0 = ncol0 + j * 16 + lg * 4;
                        gv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                        tv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (row_ok && col0 < Nseg) gv[j] = *reinterpret_cast<const float4*>(a.gateG + row * a.ldg + col0);
                        if (row_ok && col0 < Nseg) tv[j] = *reinterpret_cast<const float4*>(a.gateT + row * a.ldt + col0);
                    }
                }
#pragma unroll
                for (int j = 0; j < WCT; ++j) {
                    const int64_t col0 = ncol0 + j * 16 + lg * 4;
                    float x[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        x[r] = acc[i][j][r] + bcol[j][r];
                        if (ACT == GEOGCN_ACT_NONE || act_on) x[r] = apply_act<ACT>(x[r]);
                    }
                    if (a.accumulate) { x[0] += oldv[j].x; x[1] += oldv[j].y; x[2] += oldv[j].z; x[3] += oldv[j].w; }
                    if constexpr (GATE) {
                        x[0] = add_gate_carry(x[0], gv[j].x, tv[j].x); x[1] = add_gate_carry(x[1], gv[j].y, tv[j].y);
                        x[2] = add_gate_carry(x[2], gv[j].z, tv[j].z); x[3] = add_gate_carry(x[3], gv[j].w, tv[j].w);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (col0 + r >= Nseg) x[r] = 0.f;
                    if (row_ok && col0 < nstore) {
                        const int64_t off = row * ldc + col0;
                        if (cb16) {
                            uint2 w;
                            w.x = bf16_pack(x[0], x[1]);
                            w.y = bf16_pack(x[2], x[3]);
                            *reinterpret_cast<uint2*>((unsigned short*)Cseg + off) = w;
                        } else {
                            *reinterpret_cast<float4*>((float*)Cseg + off) = make_float4(x[0], x[1], x[2], x[3]);
                        }
                    }
                }
            }
        }
        if constexpr (KCAT != 2) __syncthreads();          // everybody done reading this tile's rows (KCAT = 2 synchronises before every load)
    }
}

// N <= 640 in one or two passes of 4 waves x 5 column tiles; K padded to exactly one of the instantiated depths
inline int rows_kp(int64_t N, int64_t K, int panel_w, int ns) {
#ifdef GEOGCN_BF16_NO_ROWS_KERNEL        // A/B build only (GEOGCN_BUILD_DEFINES): everything on gemm_bf16_kernel, as before round 3
    return 0;
#endif
    if (ns != 1 || panel_w != 0 || N > 640 || N < 1) return 0;
    const int64_t Kp = cdiv(K, BKH) * BKH;
    return (Kp == 608 || Kp == 320 || Kp == 256) ? (int)Kp : 0;
}
inline int rows_passes(int64_t N) { return N <= 4 * kRowsWCT * 16 ? 1 : 2; }

// k-concatenated launch (no activation; plain / accumulating / with the gate carry)
#ifndef GEOGCN_BF16_KCAT_MODE
#define GEOGCN_BF16_KCAT_MODE 1          // (measured, profiles/r06_bf16_kcat_mode_ab.txt: dH 1.75 ms in mode 1, 1.83 in mode 2, 1.95 as two launches) 1: both segments' tiles in LDS (one block per CU at 2 x 608), 2: one segment at a time (two blocks per CU)
#endif
template <int KP>
int launch_rows_kcat(const Bf16Args& a, hipStream_t st) {
    constexpr int KM = GEOGCN_BF16_KCAT_MODE;
    constexpr int lds = kRowsBM * ((KM == 2 ? KP / 2 : KP) * 2 + 16);
    const int passes = rows_passes(a.N);
    const int G = (int)std::min<int64_t>((int64_t)kNumCU * 2, a.n_mt);
    if (a.gateG) {
        auto kern = gemm_bf16_rows_kernel<KP, GEOGCN_ACT_NONE, true, KM>;
        static LdsAttrOnce lds_once;
        if (const int rc_ = lds_once.ensure((const void*)kern, lds)) return rc_;
        hipLaunchKernelGGL(kern, dim3((unsigned)G), dim3(TPB), lds, st, a, passes);
    } else {
        auto kern = gemm_bf16_rows_kernel<KP, GEOGCN_ACT_NONE, false, KM>;
        static LdsAttrOnce lds_once;
        if (const int rc_ = lds_once.ensure((const void*)kern, lds)) return rc_;
        hipLaunchKernelGGL(kern, dim3((unsigned)G), dim3(TPB), lds, st, a, passes);
    }
    GEOGCN_LAUNCH_CHECK("gemm_bf16_rows_kernel (k-concatenated)");
    return 0;
}

template <int KP>
int launch_rows(const Bf16Args& a, int act, hipStream_t st) {
    constexpr int lds = kRowsBM * (KP * 2 + 16);
    const int passes = rows_passes(a.N);
    const int G = (int)std::min<int64_t>((int64_t)kNumCU * 2, a.n_mt);
#define GEOGCN_R(ACT)                                                                                           \
    do {                                                                                                        \
        auto kern = gemm_bf16_rows_kernel<KP, ACT>;                                                             \
        static LdsAttrOnce lds_once;                                                                          \
        if (const int rc_ = lds_once.ensure((const void*)kern, (int)(lds))) return rc_; \
        hipLaunchKernelGGL(kern, dim3((unsigned)G), dim3(TPB), lds, st, a, passes);                             \
        GEOGCN_LAUNCH_CHECK("gemm_bf16_rows_kernel");                                                           \
    } while (0)
    if (a.gateG) {
        auto kern = gemm_bf16_rows_kernel<KP, GEOGCN_ACT_NONE, true>;
        static LdsAttrOnce lds_once;
        if (const int rc_ = lds_once.ensure((const void*)kern, (int)(lds))) return rc_;
        hipLaunchKernelGGL(kern, dim3((unsigned)G), dim3(TPB), lds, st, a, passes);
        GEOGCN_LAUNCH_CHECK("gemm_bf16_rows_kernel");
    } else if (act == GEOGCN_ACT_TANH) GEOGCN_R(GEOGCN_ACT_TANH);
    else if (act == GEOGCN_ACT_SIGMOID) GEOGCN_R(GEOGCN_ACT_SIGMOID);
    else GEOGCN_R(GEOGCN_ACT_NONE);
#undef GEOGCN_R
    return 0;
}


// ---- dW = A^T . B with bf16 products (bf16 configuration only) -------------------------------------------
// Both operands are k-STRIDED fp32 in memory ([K][M] and [K][N], K = the node dimension): a thread loads an
// 8 (k) x 4 (columns) patch as eight float4s, rounds it to bf16 and writes four 16-byte k-contiguous pieces,
// i.e. the transpose happens in registers and the LDS images are the same [row][32 k] images the MFMA
// fragments of the forward kernel read.  One 8-wave block per CU, tile BM x BN with BN spanning 256 / 320
// columns, split-K slabs in fp32 combined in fixed order by splitk_reduce (deterministic).
struct TnArgs {
    int64_t M, N, K;
    const float* A; int64_t lda;
    const float* B; int64_t ldb;
    float* W; int64_t ldw;              // slabs [nsplit][M][ldw]
    int64_t kchunk;
    int n_mt, n_nt, nsplit;
    // (round 6) a SECOND column segment of the same launch -- (dWh, dWt) = H^T . [dZ | dU]: H is read once for both.  Column tiles
    // [0, nt_per_seg) belong to B / N, the rest to B1 / N1; segment q's columns start at q * seg_w of a slab row
    int nt_per_seg = 0; const float* B1 = nullptr; int64_t ldb1 = 0; int64_t N1 = 0; int64_t seg_w = 0;
};

// PF = stages of global loads in flight per thread (register ring).  One block per CU and 20 MFMAs of 16 cycles per wave and
// stage: with ONE stage in flight (round 2) a stage took one memory latency, 1.45 us, whatever the bandwidth -- 573 stages per block
// = 0.85 ms at 600 wide for 2.1 GB of operands; three in flight let the loads of stages s+1 .. s+3 overlap.
// (the 160 x 320 tile has 100 accumulator registers: two stages in flight there, or the ring spills)
template <int BM, int BN, int PF = (BM * BN > 128 * 320 ? 2 : 3)>
__global__ __launch_bounds__(512, 1) void gemm_bf16_tn_kernel(const TnArgs a) {
    constexpr int NTH = 512, kASplit = 192;                  // threads [0,192): A patches, [192,512): B patches
    constexpr int MR = BM / 32, NR = BN / 64;                // 2 x 4 waves, wave tile (BM/2) x (BN/4)
    constexpr int kAItems = 4 * (BM / 4), kBItems = 4 * (BN / 4);
    static_assert(kAItems <= kASplit && kBItems <= NTH - kASplit, "patch lists must fit the thread ranges");
    constexpr int kImgA = BM * ROWB, kStage = (BM + BN) * ROWB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 2, wn = wid & 3, li = lane & 15, lg = lane >> 4;

    // Block b runs on XCD b % 8.  All tiles of one K slab go to the SAME XCD and start together: they read the same rows
    // of A and B, so one of them fetches a row from HBM and the others hit that XCD's L2 (with slab = b / tiles the ten
    // tiles of a slab sat on eight different XCDs and every one fetched its own copy: 7.0 GB per launch for 2.1 GB of
    // operands, profiles/r02_c_bf16_gemm.md).  Slabs beyond nsplit (grid rounded up to whole XCD rounds) do nothing.
    const int b = blockIdx.x;
    const int xcd = b % kNumXCD, s = b / kNumXCD;
    const int tiles = a.n_nt * a.n_mt;
    const int tile = s % tiles;
    const int ntile = __builtin_amdgcn_readfirstlane(tile % a.n_nt);
    const int mt = __builtin_amdgcn_readfirstlane(tile / a.n_nt);
    const int z = __builtin_amdgcn_readfirstlane(xcd + kNumXCD * (s / tiles));
    if (z >= a.nsplit) return;
    const int seg = (a.nt_per_seg && ntile >= a.nt_per_seg) ? 1 : 0;          // block-uniform
    const int nt = seg ? ntile - a.nt_per_seg : ntile;
    const int64_t Nseg = seg ? a.N1 : a.N;
    const int64_t m0 = (int64_t)mt * BM, n0 = (int64_t)nt * BN;
    const int64_t kbeg = (int64_t)z * a.kchunk, kend = min(a.K, kbeg + a.kchunk);
    const int nk = (int)((kend - kbeg + BKH - 1) / BKH);

    // my patch: operand, k8 group (8 reduction rows), c4 group (4 columns)
    const bool isA = tid < kASplit;
    const int it = isA ? tid : tid - kASplit;
    const int cols4 = isA ? BM / 4 : BN / 4;
    const bool active = it < 4 * cols4;
    const int k8 = it / cols4, c4 = it % cols4;
    const float* P = isA ? a.A : (seg ? a.B1 : a.B);
    const int64_t ld = isA ? a.lda : (seg ? a.ldb1 : a.ldb);
    const int64_t c0 = isA ? m0 : n0;
    const int64_t ctot = isA ? a.M : Nseg;
    const bool col_ok = active && (c4 * 4 < ((ctot + 3) & ~(int64_t)3) - c0);
    // (the two operands use different descriptors: built per lane group, uniform within a wave except wave 2/3
    //  boundary at thread 192 = wave 3 start, so every wave is uniform)
    float4 ring[PF][8];
    auto gload = [&](float4 (&r)[8], int kt) {
        const int64_t k0 = kbeg + (int64_t)kt * BKH;
        const __amdgpu_buffer_rsrc_t rs = tn_rsrc(P + k0 * ld + c0, ((kend - k0) * ld - c0) * 4);
        const uint32_t ld4 = (uint32_t)ld * 4u;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint32_t off = (uint32_t)(k8 * 8 + i) * ld4 + (uint32_t)c4 * 16u;
            r[i] = tn_load4(rs, col_ok ? off : kOob);
        }
    };
    auto sstore = [&](int buf, const float4 (&r)[8]) {
        if (!active) return;
        unsigned char* img = smem_raw + buf * kStage + (isA ? 0 : kImgA);
        // r[i] = row i of the patch; column e of the patch = component e of every row.  (Components are picked by name: a
        // pointer cast over the array would take its address and, with a ring of such arrays, park the ring in scratch memory)
        auto comp = [](const float4& v, int e) { return e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w; };
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            uint4 w;
            w.x = bf16_pack(comp(r[0], e), comp(r[1], e));
            w.y = bf16_pack(comp(r[2], e), comp(r[3], e));
            w.z = bf16_pack(comp(r[4], e), comp(r[5], e));
            w.w = bf16_pack(comp(r[6], e), comp(r[7], e));
            *reinterpret_cast<uint4*>(img + (c4 * 4 + e) * ROWB + k8 * 16) = w;
        }
    };

    f32x4 acc[MR][NR];
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ring slot of stage s: s % PF.  Stages past the end of the slab read as zeros (empty descriptor) and are never multiplied.
#pragma unroll
    for (int d = 0; d < PF; ++d) gload(ring[d], d);
    sstore(0, ring[0]);
    gload(ring[0], PF);
    __syncthreads();
    int cur = 0;
#pragma unroll 1
    for (int kt0 = 0; kt0 < nk; kt0 += PF) {
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const int kt = kt0 + u;          // (steps past nk in the last trip multiply and store nothing; no `break`: the ring's
        const bool live = kt < nk;       //  slots must stay compile-time constants, or the ring lands in scratch memory)
        const bool more = kt + 1 < nk;
        const unsigned char* As = smem_raw + cur * kStage;
        const unsigned char* Bs = As + kImgA;
        if (live) {
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const bf16x8 bf = *reinterpret_cast<const bf16x8*>(Bs + (wn * (BN / 4) + j * 16 + li) * ROWB + lg * 16);
#pragma unroll
            for (int i = 0; i < MR; ++i) {
                const bf16x8 af = *reinterpret_cast<const bf16x8*>(As + (wm * (BM / 2) + i * 16 + li) * ROWB + lg * 16);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf, af, acc[i][j], 0, 0, 0);   // transposed: see above
            }
        }
        }
        // stage kt + 1 was requested PF steps ago: into the other LDS image, and its slot takes stage kt + 1 + PF
        if (more) sstore(cur ^ 1, ring[(u + 1) % PF]);
        gload(ring[(u + 1) % PF], kt + 1 + PF);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        cur ^= 1;
      }
    }
    // slab z: lane (li, lg) holds C[row = li][col = 4*lg + r] of each 16x16 sub-tile
    float* Wz = a.W + (int64_t)z * a.M * a.ldw + (seg ? a.seg_w : 0);
    const int64_t n_store = (Nseg + 3) & ~(int64_t)3;
#pragma unroll
    for (int i = 0; i < MR; ++i) {
        const int64_t row = m0 + wm * (BM / 2) + i * 16 + li;
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const int64_t col0 = n0 + wn * (BN / 4) + j * 16 + lg * 4;
            float x[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) x[q] = (col0 + q < Nseg) ? acc[i][j][q] : 0.f;
            if (row < a.M && col0 < n_store)
                *reinterpret_cast<float4*>(Wz + row * a.ldw + col0) = make_float4(x[0], x[1], x[2], x[3]);
        }
    }
}

struct TnPlan {
    int bm, bn, n_mt, n_nt, nsplit;
    int64_t kchunk;
};
inline bool tn_plan(int64_t M, int64_t N, int64_t K, TnPlan& p, int n_nseg = 1) {          // (N: the wider segment)
    if (N <= 160) return false;                                   // narrow outputs stay on the fp32 kernel
    p.bm = (cdiv(M, 160) * 160 < cdiv(M, 128) * 128) ? 160 : 128;
    const int64_t w320 = cdiv(N, 320) * 320, w256 = cdiv(N, 256) * 256;
    p.bn = (w256 <= w320) ? 256 : 320;
    p.n_mt = (int)cdiv(M, p.bm);
    p.n_nt = (int)cdiv(N, p.bn) * n_nseg;
    const int64_t tiles = (int64_t)p.n_mt * p.n_nt;
    int64_t ns = std::max<int64_t>(1, kNumCU / tiles);
    if (ns >= kNumXCD) ns = ns / kNumXCD * kNumXCD;              // whole XCD rounds (the kernel groups a slab's tiles per XCD)
    ns = std::min(ns, std::max<int64_t>(1, K / (BKH * 4)));
    p.kchunk = cdiv(cdiv(K, ns), BKH) * BKH;
    p.nsplit = (int)cdiv(K, p.kchunk);
    return true;
}

}  // namespace

size_t gemm_bf16_workspace_bytes(int precision, int64_t N, int64_t K) {
    const int ns = (precision == GEOGCN_GEMM_BF16X3) ? 3 : 1;
    const int64_t Kp = cdiv(K, BKH) * BKH;
    // (the whole-rows kernel reads its weights in fragment order, padded to 4 waves x passes x 5 column tiles)
    const int64_t cols = rows_kp(N, K, 0, ns) ? std::max<int64_t>(N, (int64_t)4 * rows_passes(N) * kRowsWCT * 16) : N;
    return (size_t)ns * (size_t)cols * (size_t)Kp * sizeof(unsigned short);
}

// ---- the highway block's forward pair in ONE launch (bf16 configuration): Z = H . Wh as bf16 (the SpMM's operand) or fp32,
// T = act1(H . Wt + bias1) as fp32 -- H read from HBM and rounded to bf16 ONCE for both (whole-rows kernel, two column
// segments); widths the kernel does not take run as the two separate launches.  Same arithmetic either way: bit-identical.
static size_t dual_slot_bytes(int64_t N, int kp) { return (size_t)4 * rows_passes(N) * kRowsWCT * (kp / BKH) * 1024; }
static int dual_kp(int64_t N0, int64_t N1, int64_t K) {
    const int k0 = rows_kp(N0, K, 0, 1), k1 = rows_kp(N1, K, 0, 1);
    return (k0 && k0 == k1) ? k0 : 0;
}
size_t gemm_bf16_dual_workspace_bytes(int64_t N0, int64_t N1, int64_t K) {
    const size_t separate = std::max(gemm_bf16_workspace_bytes(GEOGCN_GEMM_BF16, N0, K), gemm_bf16_workspace_bytes(GEOGCN_GEMM_BF16, N1, K));
    const int kp = dual_kp(N0, N1, K);
    return kp ? std::max(separate, dual_slot_bytes(N0, kp) + dual_slot_bytes(N1, kp)) : separate;
}
int gemm_bf16_dual_dispatch(int64_t M, int64_t N0, int64_t N1, int64_t K, const float* A, int64_t lda, const float* B0,
                            int64_t ldb0, const float* B1, int64_t ldb1, void* C0, int64_t ldc0, int c0_bf16, float* C1,
                            int64_t ldc1, const float* bias1, int act1, void* ws, size_t ws_bytes, hipStream_t st) {
    const size_t need = gemm_bf16_dual_workspace_bytes(N0, N1, K);
    GEOGCN_REQUIRE(ws && ws_bytes >= need && aligned16(ws), GEOGCN_E_ARG, "gemm_dual_bf16: workspace too small (%zu < %zu)", ws_bytes, need);
    const int kp = dual_kp(N0, N1, K);
    if (!kp || lda % 4 != 0) {
        if (const int rc = gemm_bf16_dispatch(GEOGCN_GEMM_BF16, 0, M, N0, K, A, lda, B0, ldb0, C0, ldc0, c0_bf16, nullptr, GEOGCN_ACT_NONE, 0,
                                              ws, ws_bytes, st, 0, 0, nullptr))
            return rc;
        return gemm_bf16_dispatch(GEOGCN_GEMM_BF16, 0, M, N1, K, A, lda, B1, ldb1, C1, ldc1, 0, bias1, act1, 0, ws, ws_bytes, st, 0, 0, nullptr);
    }
    const int nks = kp / BKH;
    unsigned short* p0 = (unsigned short*)ws;
    unsigned short* p1 = p0 + dual_slot_bytes(N0, kp) / sizeof(unsigned short);
    const int t0 = 4 * rows_passes(N0) * kRowsWCT, t1 = 4 * rows_passes(N1) * kRowsWCT;
    hipLaunchKernelGGL((prep_b_frag_kernel<1>), dim3((unsigned)std::min<int64_t>(cdiv((int64_t)t0 * nks * 512, TPB), 1024)), dim3(TPB), 0, st,
                       B0, ldb0, (int)K, (int)N0, nks, t0, 0, p0);
    hipLaunchKernelGGL((prep_b_frag_kernel<1>), dim3((unsigned)std::min<int64_t>(cdiv((int64_t)t1 * nks * 512, TPB), 1024)), dim3(TPB), 0, st,
                       B1, ldb1, (int)K, (int)N1, nks, t1, 0, p1);
    GEOGCN_LAUNCH_CHECK("prep_b_frag_kernel");
    Bf16Args a{M, N0, K, A, lda, p0, kp, C0, ldc0, nullptr, 0, (int)cdiv(M, kRowsBM), 1, c0_bf16,
               c0_bf16 ? ((N0 + 7) & ~(int64_t)7) : ((N0 + 3) & ~(int64_t)3), 0, 0};
    a.n_nseg = 2;
    a.passes0 = rows_passes(N0); a.passes1 = rows_passes(N1);
    a.Bp1 = p1; a.C1 = C1; a.ldc1 = ldc1; a.bias1 = bias1; a.N1 = N1; a.c_bf16_1 = 0; a.n_store1 = (N1 + 3) & ~(int64_t)3;
    if (kp == 608) return launch_rows<608>(a, act1, st);
    if (kp == 320) return launch_rows<320>(a, act1, st);
    return launch_rows<256>(a, act1, st);
}

// ---- (round 6) dH = dZ . Wh^T + dU . Wt^T of the highway block in ONE launch in the bf16 configuration: both products into one
// accumulator, one pass over dH (until now: the first product written -- with the carry -- and the second accumulated onto it: dH written,
// read and written again, 2.1 GB per 600-wide block at the TwitterUS size).  Shapes: both K pad to the same 608 / 320 / 256, N <= 640.
static int kcat_kp(int64_t N, int64_t K0, int64_t K1) {
    const int k0 = rows_kp(N, K0, 0, 1), k1 = rows_kp(N, K1, 0, 1);
    return (k0 && k0 == k1) ? k0 : 0;
}
size_t gemm_bf16_kcat_workspace_bytes(int64_t N, int64_t K0, int64_t K1) {
    const size_t separate = std::max(gemm_bf16_workspace_bytes(GEOGCN_GEMM_BF16, N, K0), gemm_bf16_workspace_bytes(GEOGCN_GEMM_BF16, N, K1));
    const int kp = kcat_kp(N, K0, K1);
    return kp ? std::max(separate, 2 * dual_slot_bytes(N, kp)) : separate;
}
bool gemm_bf16_kcat_native(int64_t N, int64_t K0, int64_t K1) { return kcat_kp(N, K0, K1) != 0; }
// returns 1 when the shape is not one the k-concatenated kernel takes (the caller then runs two launches)
int gemm_bf16_kcat_dispatch(int transB, int64_t M, int64_t N, int64_t K0, int64_t K1, const float* A0, int64_t lda0, const float* B0,
                            int64_t ldb0, const float* A1, int64_t lda1, const float* B1, int64_t ldb1, float* C, int64_t ldc,
                            int accumulate, void* ws, size_t ws_bytes, hipStream_t st, const GateOps* gate) {
    const int kp = kcat_kp(N, K0, K1);
    if (!kp || lda0 % 4 != 0 || lda1 % 4 != 0) return 1;
    const size_t need = 2 * dual_slot_bytes(N, kp);
    GEOGCN_REQUIRE(ws && ws_bytes >= need && aligned16(ws), GEOGCN_E_ARG, "gemm_kcat(bf16): workspace too small (%zu < %zu)", ws_bytes, need);
    const int nks = kp / BKH, n_tiles = 4 * rows_passes(N) * kRowsWCT;
    unsigned short* planes = (unsigned short*)ws;
    const unsigned fgrid = (unsigned)std::min<int64_t>(cdiv((int64_t)n_tiles * nks * 512, TPB), 1024);
    hipLaunchKernelGGL((prep_b_frag_kernel<1>), dim3(fgrid), dim3(TPB), 0, st, B0, ldb0, (int)K0, (int)N, nks, n_tiles, transB, planes, 0, 2 * nks);
    hipLaunchKernelGGL((prep_b_frag_kernel<1>), dim3(fgrid), dim3(TPB), 0, st, B1, ldb1, (int)K1, (int)N, nks, n_tiles, transB, planes, nks, 2 * nks);
    GEOGCN_LAUNCH_CHECK("prep_b_frag_kernel");
    Bf16Args a{M, N, K0, A0, lda0, planes, 2 * kp, C, ldc, nullptr, accumulate, (int)cdiv(M, kRowsBM), 1, 0, (N + 3) & ~(int64_t)3, 0, 0};
    a.A1 = A1; a.lda1 = lda1; a.K1 = K1;
    if (gate) { a.gateG = gate->G; a.ldg = gate->ldg; a.gateT = gate->T; a.ldt = gate->ldt; }
    if (kp == 608) return launch_rows_kcat<1216>(a, st);
    if (kp == 320) return launch_rows_kcat<640>(a, st);
    return launch_rows_kcat<512>(a, st);
}

// called by geogcn_gemm_f32 for transA == 0 and precision != F32
int gemm_bf16_dispatch(int precision, int transB, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
                       const float* B, int64_t ldb, void* C, int64_t ldc, int c_bf16, const float* bias, int act,
                       int accumulate, void* ws, size_t ws_bytes, hipStream_t st, int panel_w, int64_t panel_R, const GateOps* gate) {
    const int ns = (precision == GEOGCN_GEMM_BF16X3) ? 3 : 1;
    const int Kp = (int)(cdiv(K, BKH) * BKH);
    const size_t need = gemm_bf16_workspace_bytes(precision, N, K);
    GEOGCN_REQUIRE(ws && ws_bytes >= need && aligned16(ws), GEOGCN_E_ARG, "gemm_f32(bf16): workspace too small (%zu < %zu)",
                   ws_bytes, need);
    unsigned short* planes = (unsigned short*)ws;
    if (const int kp = rows_kp(N, K, panel_w, ns); kp && lda % 4 == 0) {
        const int n_tiles = 4 * rows_passes(N) * kRowsWCT, nks = kp / BKH;
        const unsigned fgrid = (unsigned)std::min<int64_t>(cdiv((int64_t)n_tiles * nks * 512, TPB), 1024);
        hipLaunchKernelGGL((prep_b_frag_kernel<1>), dim3(fgrid), dim3(TPB), 0, st, B, ldb, (int)K, (int)N, nks, n_tiles, transB, planes);
        GEOGCN_LAUNCH_CHECK("prep_b_frag_kernel");
        Bf16Args a{M, N, K, A, lda, planes, kp, C, ldc, bias, accumulate, (int)cdiv(M, kRowsBM), 1, c_bf16,
                   c_bf16 ? ((N + 7) & ~(int64_t)7) : ((N + 3) & ~(int64_t)3), 0, 0};
        if (gate) { a.gateG = gate->G; a.ldg = gate->ldg; a.gateT = gate->T; a.ldt = gate->ldt; }
        if (kp == 608) return launch_rows<608>(a, act, st);
        if (kp == 320) return launch_rows<320>(a, act, st);
        return launch_rows<256>(a, act, st);
    }
    if (gate) {          // not a whole-rows shape: the carry written first, the product accumulated onto it (same values, same order)
        if (const int rc = geogcn_gate_carry_f32(M, (int32_t)N, gate->G, gate->ldg, gate->T, gate->ldt, (float*)C, ldc, (void*)st)) return rc;
        accumulate = 1;
    }
    const unsigned pgrid = (unsigned)std::min<int64_t>(cdiv((int64_t)N * Kp, TPB), 1024);
    if (ns == 3)
        hipLaunchKernelGGL((prep_b_planes_kernel<3>), dim3(pgrid), dim3(TPB), 0, st, B, ldb, (int)K, (int)N, Kp, transB, planes);
    else
        hipLaunchKernelGGL((prep_b_planes_kernel<1>), dim3(pgrid), dim3(TPB), 0, st, B, ldb, (int)K, (int)N, Kp, transB, planes);
    GEOGCN_LAUNCH_CHECK("prep_b_planes_kernel");
    // (ties -> 160: fewer N tiles, and every N tile converts its A tile from fp32 again)
    const int bn = (cdiv(N, 160) * 160 <= cdiv(N, 128) * 128) ? 160 : 128;
    Bf16Args a{M, N, K, A, lda, planes, Kp, C, ldc, bias, accumulate, (int)cdiv(M, 128), (int)cdiv(N, bn), c_bf16,
               c_bf16 ? ((N + 7) & ~(int64_t)7) : ((N + 3) & ~(int64_t)3), panel_w, panel_R};
    if (ns == 3) {      // 8 waves per block: half the accumulators / staging registers per lane
        if (bn == 160) return launch_bf16<128, 160, 3, 512>(a, act, st);
        return launch_bf16<128, 128, 3, 512>(a, act, st);
    }
#ifdef GEOGCN_BF16_PROBE_BUILD      // ablation build only (hipcc -DGEOGCN_BF16_PROBE_BUILD; profiles/r02_c_bf16_gemm.md)
    static const int probe = [] { const char* e = getenv("GEOGCN_BF16_PROBE"); return e ? atoi(e) : 0; }();
    if (probe && bn == 160 && act == GEOGCN_ACT_NONE) {
        using Cfg = BCfg<128, 160, 1, 256>;
        const int G = (int)std::min<int64_t>((int64_t)kNumCU * 2, cdiv((int64_t)a.n_mt * a.n_nt, kNumXCD) * kNumXCD);
#define GEOGCN_P(P_)                                                                                                  \
    do {                                                                                                              \
        auto kern = gemm_bf16_kernel<128, 160, 1, GEOGCN_ACT_NONE, 256, P_>;                                          \
        GEOGCN_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::kLdsBytes)); \
        hipLaunchKernelGGL(kern, dim3((unsigned)G), dim3(256), Cfg::kLdsBytes, st, a);                              \
    } while (0)
        switch (probe) {
            case 1: GEOGCN_P(1); break;
            case 2: GEOGCN_P(2); break;
            case 3: GEOGCN_P(3); break;
            case 4: GEOGCN_P(4); break;
            case 7: GEOGCN_P(7); break;
            case 8: GEOGCN_P(8); break;
            case 15: GEOGCN_P(15); break;
            default: break;
        }
#undef GEOGCN_P
        if (probe == 1 || probe == 2 || probe == 3 || probe == 4 || probe == 7 || probe == 8 || probe == 15) return 0;
    }
#endif
    if (bn == 160) return launch_bf16<128, 160, 1, 256>(a, act, st);
    return launch_bf16<128, 128, 1, 256>(a, act, st);
}


size_t gemm_bf16_tn_workspace_bytes(int64_t M, int64_t N, int64_t K) {
    TnPlan p;
    if (!tn_plan(M, N, K, p)) return 0;
    return (size_t)p.nsplit * (size_t)M * (size_t)((N + 3) & ~(int64_t)3) * sizeof(float);
}

// (round 6) (dW0, dW1) = A^T . [B0 | B1] in one launch (the highway block's two weight gradients: H read once); returns 1 if not handled
size_t gemm_bf16_tn_dual_workspace_bytes(int64_t M, int64_t N0, int64_t N1, int64_t K) {
    TnPlan p;
    const size_t separate = std::max(gemm_bf16_tn_workspace_bytes(M, N0, K), gemm_bf16_tn_workspace_bytes(M, N1, K));
    if (!tn_plan(M, std::max(N0, N1), K, p, 2) || std::min(N0, N1) <= 160) return separate;
    const int64_t seg_w = (std::max(N0, N1) + 3) & ~(int64_t)3;
    return std::max(separate, (size_t)p.nsplit * (size_t)M * (size_t)(2 * seg_w) * sizeof(float));
}
int gemm_bf16_tn_dual_dispatch(int64_t M, int64_t N0, int64_t N1, int64_t K, const float* A, int64_t lda, const float* B0, int64_t ldb0,
                               const float* B1, int64_t ldb1, float* C0, int64_t ldc0, float* C1, int64_t ldc1, void* ws, size_t ws_bytes,
                               hipStream_t st) {
    TnPlan p;
    if (!tn_plan(M, std::max(N0, N1), K, p, 2) || std::min(N0, N1) <= 160) return 1;
    const int64_t seg_w = (std::max(N0, N1) + 3) & ~(int64_t)3, ldw = 2 * seg_w;
    const size_t need = (size_t)p.nsplit * (size_t)M * (size_t)ldw * sizeof(float);
    GEOGCN_REQUIRE(ws && ws_bytes >= need && aligned16(ws), GEOGCN_E_ARG, "gemm_dual(bf16, transA): workspace too small (%zu < %zu)", ws_bytes, need);
    TnArgs a{M, N0, K, A, lda, B0, ldb0, (float*)ws, ldw, p.kchunk, p.n_mt, p.n_nt, p.nsplit};
    a.nt_per_seg = p.n_nt / 2; a.B1 = B1; a.ldb1 = ldb1; a.N1 = N1; a.seg_w = seg_w;
    const dim3 grid((unsigned)((int64_t)p.n_mt * p.n_nt * cdiv(p.nsplit, kNumXCD) * kNumXCD));
#define GEOGCN_TN(BM_, BN_)                                                                                     \
    do {                                                                                                        \
        auto kern = gemm_bf16_tn_kernel<BM_, BN_>;                                                              \
        constexpr int lds = 2 * (BM_ + BN_) * ROWB;                                                             \
        static LdsAttrOnce lds_once;                                                                            \
        if (const int rc_ = lds_once.ensure((const void*)kern, (int)(lds))) return rc_;                         \
        hipLaunchKernelGGL(kern, grid, dim3(512), lds, st, a);                                                  \
        GEOGCN_LAUNCH_CHECK("gemm_bf16_tn_kernel");                                                             \
    } while (0)
    if (p.bm == 160 && p.bn == 320) GEOGCN_TN(160, 320);
    else if (p.bm == 160) GEOGCN_TN(160, 256);
    else if (p.bn == 320) GEOGCN_TN(128, 320);
    else GEOGCN_TN(128, 256);
#undef GEOGCN_TN
    if (const int rc = splitk_reduce_launch(M, N0, p.nsplit, (const float*)ws, ldw, C0, ldc0, nullptr, GEOGCN_ACT_NONE, 0, st)) return rc;
    return splitk_reduce_launch(M, N1, p.nsplit, (const float*)ws + seg_w, ldw, C1, ldc1, nullptr, GEOGCN_ACT_NONE, 0, st);
}

// dW = A^T . B (A: K x M, B: K x N, fp32 in memory) with bf16 products; returns 1 if this shape is not handled
// (the caller then runs the fp32 kernel), 0 on success, an error code otherwise
int gemm_bf16_tn_dispatch(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B, int64_t ldb,
                          float* C, int64_t ldc, const float* bias, int act, int accumulate, void* ws, size_t ws_bytes,
                          hipStream_t st) {
    TnPlan p;
    if (!tn_plan(M, N, K, p)) return 1;
    const size_t need = gemm_bf16_tn_workspace_bytes(M, N, K);
    GEOGCN_REQUIRE(ws && ws_bytes >= need && aligned16(ws), GEOGCN_E_ARG, "gemm_f32(bf16, transA): workspace too small (%zu < %zu)",
                   ws_bytes, need);
    const int64_t ldw = (N + 3) & ~(int64_t)3;
    TnArgs a{M, N, K, A, lda, B, ldb, (float*)ws, ldw, p.kchunk, p.n_mt, p.n_nt, p.nsplit};
    const dim3 grid((unsigned)((int64_t)p.n_mt * p.n_nt * cdiv(p.nsplit, kNumXCD) * kNumXCD));
#define GEOGCN_TN(BM_, BN_)                                                                                     \
    do {                                                                                                        \
        auto kern = gemm_bf16_tn_kernel<BM_, BN_>;                                                              \
        constexpr int lds = 2 * (BM_ + BN_) * ROWB;                                                             \
        static LdsAttrOnce lds_once;                                                                          \
        if (const int rc_ = lds_once.ensure((const void*)kern, (int)(lds))) return rc_; \
        hipLaunchKernelGGL(kern, grid, dim3(512), lds, st, a);                                                  \
        GEOGCN_LAUNCH_CHECK("gemm_bf16_tn_kernel");                                                             \
    } while (0)
    if (p.bm == 160 && p.bn == 320) GEOGCN_TN(160, 320);
    else if (p.bm == 160) GEOGCN_TN(160, 256);
    else if (p.bn == 320) GEOGCN_TN(128, 320);
    else GEOGCN_TN(128, 256);
#undef GEOGCN_TN
    return splitk_reduce_launch(M, N, p.nsplit, (const float*)ws, ldw, C, ldc, bias, act, accumulate, st);
}

}  // namespace geogcn
