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
#include "gemm_call.h"

#include <stdlib.h>

#include <algorithm>

namespace geogcn {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 32;
constexpr int KPITCH = BK + 4;     // k-contiguous image pitch (floats): 144 B rows, 16-B aligned
constexpr int TPB = 256;

template <int R, int NTH>
struct KContig {                    // R rows x BK floats, row-major, pitch KPITCH
    static constexpr int kFloats = R * KPITCH;
    static constexpr int kRowsPerPass = NTH / 8;    // NTH threads x float4 = NTH/8 rows per pass
    static constexpr int kIters = (R + kRowsPerPass - 1) / kRowsPerPass;
    static constexpr bool kExact = (R % kRowsPerPass) == 0;
};
template <int Ccols, int NTH>
struct KStrided {                   // BK rows x Ccols floats, pitch Ccols + 4
    static constexpr int kPitch = Ccols + 4;
    static constexpr int kFloats = BK * kPitch;
    static constexpr int kF4PerRow = Ccols / 4;
    static constexpr int kTotal = BK * kF4PerRow;
    static constexpr int kIters = (kTotal + NTH - 1) / NTH;
    static constexpr bool kExact = (kTotal % NTH) == 0;
};

template <int R, int NTH>
using KContigRegs = float4[KContig<R, NTH>::kIters];
template <int Ccols, int NTH>
using KStridedRegs = float4[KStrided<Ccols, NTH>::kIters];

// ---- global -> registers ----------------------------------------------------------------------
// Buffer loads (raw, stride 0): the descriptor is rebased on the tile / stage origin, so that
//   * rows past the end of the operand (M tail, K tail of a k-strided operand, end of a split-K chunk) lie
//     beyond num_records and read as 0 in hardware -- no exec-masked branches, straight-line code, and the
//     compiler can count its vmcnt waits exactly;
//   * the per-lane offset is a loop-invariant 32-bit VGPR (no 64-bit address arithmetic per load);
//   * operands larger than 4 GB work: only the distance from the tile origin must fit 31 bits.
// The contiguous dimension is guarded by swapping the offset for one that is out of range: its limit is
// rounded UP to a multiple of 4 (<= ld), so a float4 is wholly inside or wholly outside; elements in
// [lim, lim4) are pad columns, which are zero by the geogcn.h convention.
constexpr uint32_t kOobOffset = 0x80000000u;     // > any num_records we ever set (clamped to 2^31 - 1)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t tile_rsrc(const float* base, int64_t bytes) {
    // base and bytes depend on blockIdx and loop counters only; readfirstlane states that uniformity for the
    // compiler (a descriptor it cannot prove uniform is loaded through a per-lane "waterfall" loop)
    const uint32_t n = bytes <= 0 ? 0u : (bytes > 0x7FFFFFFFll ? 0x7FFFFFFFu : (uint32_t)bytes);
    const uint64_t b = reinterpret_cast<uint64_t>(base);
    // (readfirstlane returns int: go through uint32_t, or the low word is sign-extended into the high one)
    const uint64_t bu = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(b >> 32)) << 32) |
                        (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)b);
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(bu), 0, __builtin_amdgcn_readfirstlane(n), 0x00020000);
}
__device__ __forceinline__ float4 buffer_load4(__amdgpu_buffer_rsrc_t r, uint32_t off) {
    const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0));
    return make_float4(v.x, v.y, v.z, v.w);
}

// k-contiguous operand: memory [R_total][K] row-major (ld), tile rows r0.., k range k0..k0+31
template <int R, int NTH>
__device__ __forceinline__ void gload_kcontig(KContigRegs<R, NTH>& reg, const float* __restrict__ P,
                                              int64_t ld, int64_t r0, int64_t Rtot, int64_t k0, int64_t Kend) {
    using L = KContig<R, NTH>;
    const int tid = threadIdx.x;
    const int f4 = tid & 7;
    const int rr = tid >> 3;
    const __amdgpu_buffer_rsrc_t rs = tile_rsrc(P + r0 * ld + k0, ((Rtot - r0) * ld - k0) * 4);
    const bool k_ok = k0 + f4 * 4 < ((Kend + 3) & ~(int64_t)3);
    const uint32_t ld4 = (uint32_t)ld * 4u;
#pragma unroll
    for (int i = 0; i < L::kIters; ++i) {
        const int row = rr + L::kRowsPerPass * i;
        const uint32_t off = (uint32_t)row * ld4 + (uint32_t)f4 * 16u;
        const bool ok = k_ok && (L::kExact || row < R);
        reg[i] = buffer_load4(rs, ok ? off : kOobOffset);
    }
}
template <int R, int NTH>
__device__ __forceinline__ void sstore_kcontig(float* __restrict__ S, const KContigRegs<R, NTH>& reg) {
    using L = KContig<R, NTH>;
    const int tid = threadIdx.x;
    const int f4 = tid & 7;
    const int rr = tid >> 3;
#pragma unroll
    for (int i = 0; i < L::kIters; ++i) {
        const int row = rr + L::kRowsPerPass * i;
        if (L::kExact || row < R) *reinterpret_cast<float4*>(S + row * KPITCH + f4 * 4) = reg[i];
    }
}
// k-strided operand: memory [K][C_total] row-major (ld), tile cols c0.., k range k0..k0+31
template <int Ccols, int NTH>
__device__ __forceinline__ void gload_kstrided(KStridedRegs<Ccols, NTH>& reg, const float* __restrict__ P,
                                               int64_t ld, int64_t c0, int64_t Ctot, int64_t k0, int64_t Kend) {
    using L = KStrided<Ccols, NTH>;
    const int tid = threadIdx.x;
    constexpr int F4R = L::kF4PerRow;
    const __amdgpu_buffer_rsrc_t rs = tile_rsrc(P + k0 * ld + c0, ((Kend - k0) * ld - c0) * 4);
    const int64_t c_lim = ((Ctot + 3) & ~(int64_t)3) - c0;       // valid columns of this tile (multiple of 4)
    const uint32_t ld4 = (uint32_t)ld * 4u;
#pragma unroll
    for (int i = 0; i < L::kIters; ++i) {
        const int e = tid + NTH * i;
        const int kr = e / F4R;
        const int c4 = e % F4R;
        const uint32_t off = (uint32_t)kr * ld4 + (uint32_t)c4 * 16u;
        const bool ok = c4 * 4 < c_lim && (L::kExact || e < L::kTotal);
        reg[i] = buffer_load4(rs, ok ? off : kOobOffset);
    }
}
template <int Ccols, int NTH>
__device__ __forceinline__ void sstore_kstrided(float* __restrict__ S, const KStridedRegs<Ccols, NTH>& reg) {
    using L = KStrided<Ccols, NTH>;
    const int tid = threadIdx.x;
    constexpr int F4R = L::kF4PerRow;
#pragma unroll
    for (int i = 0; i < L::kIters; ++i) {
        const int e = tid + NTH * i;
        const int kr = e / F4R;
        const int c4 = e % F4R;
        if (L::kExact || e < L::kTotal) *reinterpret_cast<float4*>(S + kr * L::kPitch + c4 * 4) = reg[i];
    }
}

// WM x WN waves per block (2 x 2 = 256 threads, two blocks per CU; 2 x 4 = 512 threads, one block per CU:
// the same two waves per SIMD, but one tile spans all of N <= 320, so the streamed operand is staged once)
template <int BM, int BN, bool AT, bool BT, int WM = 2, int WN = 2>
struct GemmCfg {
    static constexpr int NTH = 64 * WM * WN;
    static constexpr int kAFloats = AT ? KStrided<BM, NTH>::kFloats : KContig<BM, NTH>::kFloats;
    static constexpr int kBFloats = BT ? KContig<BN, NTH>::kFloats : KStrided<BN, NTH>::kFloats;
    static constexpr int kAIters = AT ? KStrided<BM, NTH>::kIters : KContig<BM, NTH>::kIters;
    static constexpr int kBIters = BT ? KContig<BN, NTH>::kIters : KStrided<BN, NTH>::kIters;
    static constexpr int kStageFloats = kAFloats + kBFloats;
    static constexpr size_t kLdsBytes = 2 * (size_t)kStageFloats * sizeof(float);
    static constexpr int kWaveM = BM / WM, kWaveN = BN / WN;     // wave tile
    static constexpr int MR = kWaveM / 16;   // 16x16 tiles per wave along M
    static constexpr int NR = kWaveN / 16;
    static constexpr int kBlocksPerCU = (NTH == 256 && 2 * kLdsBytes <= 160 * 1024) ? 2 : 1;
    static_assert(kWaveM % 16 == 0 && kWaveN % 16 == 0, "wave tile must be a multiple of the 16x16 MFMA");
};

// One launch multiplies up to two SEGMENTS (gcnmodel.py:281-286: the highway block's conv branch and gate read the
// same input; their backward adds two products into the same dH):
//   n_nseg = 2  "dual":  C[q] = act_q(op(A[0]) . B[q] + bias[q]), q = 0, 1 -- the A tile is staged for both weights by the
//               same XCD (shared through its L2), tile column nt belongs to segment nt / nt_per_seg;
//   n_kseg = 2  "k-concatenated":  C[0] = A[0].op(B[0]) + A[1].op(B[1]) [+ C[0]] -- one accumulator, one pass over C.
// Never both; split-K (transA) only with n_kseg = 1.
struct GemmArgs {
    int64_t M;
    const float* A[2]; int64_t lda[2];
    const float* B[2]; int64_t ldb[2];
    float* C[2]; int64_t ldc[2];
    const float* bias[2];
    int64_t N[2];              // output columns of N segment q
    int64_t K[2];              // reduction length of K segment q
    int act_on[2];             // apply the kernel's ACT to N segment q (0 = linear)
    int accumulate;
    int64_t kchunk;
    int n_mt, n_nt, nt_per_seg, n_split, xcd_order, n_kseg, nk0;
    int64_t slab_seg_w;        // MODE 1: column offset of N segment 1 inside a slab row
    // MODE 0, N segment 0 only: C laid out as feature panels [N / panel_w][panel_R][panel_w] instead of row-major
    // (element (i, j) at ((j / panel_w) * panel_R + i) * panel_w + j % panel_w): the send buffer of the multi-GPU
    // repartition all-to-all, written straight from the accumulators.  0 = row-major.
    int panel_w;
    int64_t panel_R;
};

// Tile coordinates are wave-uniform (functions of blockIdx and loop counters); readfirstlane keeps them in
// SGPRs so that everything derived from them (buffer descriptors, loop control) is scalar code.
struct TileCoord {
    int mt, nt, z, nk;      // nk == 0: past the end of this block's list;  nt = tile column INSIDE its segment
    int seg;                // N segment
};
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// Persistent blocks: block p walks its tile list p, p+G, ... (XCD-aware: the blocks of one XCD walk the
// N tiles / K slices of the same M tiles concurrently, so the A panel is shared through that XCD's L2).
__device__ __forceinline__ TileCoord decode_tile(const GemmArgs& a, int p, int G, int j) {
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
    TileCoord t;
    t.mt = uni(mt);
    t.seg = uni(nt / a.nt_per_seg);
    t.nt = uni(nt % a.nt_per_seg);
    t.z = uni(z);
    int nk;
    if (a.n_kseg == 2) {
        nk = a.nk0 + (int)((a.K[1] + BK - 1) / BK);
    } else {
        const int64_t kbeg = (int64_t)t.z * a.kchunk;
        const int64_t kend = min(a.K[0], kbeg + a.kchunk);
        nk = (int)((kend - kbeg + BK - 1) / BK);
    }
    if (t.mt >= a.n_mt) nk = 0;
    t.nk = uni(nk < 0 ? 0 : nk);
    return t;
}

// MODE 0: C = act(acc + bias) [+ C if accumulate];  MODE 1: split-K slab z (raw partial sums)
// The k-loop is FLATTENED across the block's tiles: the global loads of stage s+2 are always in flight
// during the MFMAs of stage s, also across a tile boundary, so the short K = 300 contractions of the GCN
// (10 stages per tile) pay no per-tile prologue.  The loop body is branch-free apart from wave-uniform
// control (K tail, epilogue, tile advance): loads past the end of the list use an empty buffer descriptor
// (they return zeros) and their LDS image is written but never multiplied, so the compiler can count its
// vmcnt waits exactly instead of draining the memory pipeline at control-flow joins.
template <int BM, int BN, bool AT, bool BT, int ACT, int MODE, int PROBE = 0, int WM = 2, int WN = 2>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN == 4) ? 2 : 1) void gemm_kernel(const GemmArgs a) {
    // PROBE (tools/micro/gemm_variants.hip only; 0 in the library): 1 = no global loads, 2 = no C stores,
    // 4 = no K-tail early-out, 8 = no LDS stores, 16 = s_setprio around the MFMA section
    using Cfg = GemmCfg<BM, BN, AT, BT, WM, WN>;
    constexpr int NTH = Cfg::NTH;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int p = blockIdx.x, G = gridDim.x;

    const int lane = threadIdx.x & 63;
    const int wid = threadIdx.x >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int li = lane & 15, lg = lane >> 4;

    f32x4 acc[Cfg::MR][Cfg::NR];
#pragma unroll
    for (int i = 0; i < Cfg::MR; ++i)
#pragma unroll
        for (int jn = 0; jn < Cfg::NR; ++jn) acc[i][jn] = f32x4{0.f, 0.f, 0.f, 0.f};

    // two register sets: while stage s is multiplied, stage s+1 waits in one set to be written to LDS and
    // the global loads of stage s+2 land in the other (two stages of HBM latency tolerance)
    float4 ra0[Cfg::kAIters], rb0[Cfg::kBIters], ra1[Cfg::kAIters], rb1[Cfg::kBIters];
    // reduction window of stage kt of tile t: K segment, first k, end of the window (all wave-uniform)
    struct KWin { int ks; int64_t k0, kend; };
    auto kwin = [&](const TileCoord& t, int kt) -> KWin {
        KWin w;
        if (a.n_kseg == 2) {
            w.ks = uni(kt >= a.nk0 ? 1 : 0);
            w.k0 = (int64_t)(w.ks ? kt - a.nk0 : kt) * BK;
            w.kend = w.ks ? a.K[1] : a.K[0];
        } else {
            const int64_t kbeg = (int64_t)t.z * a.kchunk;
            w.ks = 0;
            w.k0 = kbeg + (int64_t)kt * BK;
            w.kend = min(a.K[0], kbeg + a.kchunk);
        }
        return w;
    };
    auto gload = [&](float4 (&ra)[Cfg::kAIters], float4 (&rb)[Cfg::kBIters], const TileCoord& t, int kt) {
        const KWin w = kwin(t, kt);
        // past the end of the list (nk == 0): kend = k0 makes every descriptor empty
        const int64_t k0 = w.k0;
        const int64_t kend = t.nk > 0 ? w.kend : k0;
        const int64_t m0 = (int64_t)t.mt * BM, n0 = (int64_t)t.nt * BN;
        const int sb = w.ks | t.seg;             // which B (at most one of the two indices is non-zero)
        const float* Ap = w.ks ? a.A[1] : a.A[0];
        const int64_t lda = w.ks ? a.lda[1] : a.lda[0];
        const float* Bp = sb ? a.B[1] : a.B[0];
        const int64_t ldb = sb ? a.ldb[1] : a.ldb[0];
        const int64_t Mlim = t.nk > 0 ? a.M : 0, Nlim = t.nk > 0 ? (t.seg ? a.N[1] : a.N[0]) : 0;
        if constexpr (AT) gload_kstrided<BM, NTH>(ra, Ap, lda, m0, Mlim, k0, kend);
        else gload_kcontig<BM, NTH>(ra, Ap, lda, m0, Mlim, k0, kend);
        if constexpr (BT) gload_kcontig<BN, NTH>(rb, Bp, ldb, n0, Nlim, k0, kend);
        else gload_kstrided<BN, NTH>(rb, Bp, ldb, n0, Nlim, k0, kend);
    };
    auto sstore = [&](int buf, const float4 (&ra)[Cfg::kAIters], const float4 (&rb)[Cfg::kBIters]) {
        float* As = smem + buf * Cfg::kStageFloats;
        float* Bs = As + Cfg::kAFloats;
        if constexpr (AT) sstore_kstrided<BM, NTH>(As, ra);
        else sstore_kcontig<BM, NTH>(As, ra);
        if constexpr (BT) sstore_kcontig<BN, NTH>(Bs, rb);
        else sstore_kstrided<BN, NTH>(Bs, rb);
    };

    int cj = 0, ckt = 0;                     // compute cursor (tile index in my list, stage)
    TileCoord ct = decode_tile(a, p, G, 0);
    if (ct.nk == 0) return;
    int lj = 0, lkt = 0;                     // load cursor
    TileCoord lt = ct;
    auto advance_load = [&]() {
        if (lt.nk == 0) return;
        if (++lkt == lt.nk) { lt = decode_tile(a, p, G, ++lj); lkt = 0; }
    };
    gload(ra0, rb0, lt, 0);
    sstore(0, ra0, rb0);
    advance_load();
    gload(ra1, rb1, lt, lkt);                // stage 1 (zeros if there is none)
    advance_load();
    __syncthreads();
    int cur = 0;
    // one pipeline step; (la, lb) = set to load stage s+2 into, (sa, sb) = set holding stage s+1
    auto step = [&](float4 (&la)[Cfg::kAIters], float4 (&lb)[Cfg::kBIters], const float4 (&sa)[Cfg::kAIters],
                    const float4 (&sb)[Cfg::kBIters]) {
        if (!(PROBE & 1)) gload(la, lb, lt, lkt);
        advance_load();
        const float* As = smem + cur * Cfg::kStageFloats;
        const float* Bs = As + Cfg::kAFloats;
        const KWin cw = kwin(ct, ckt);
        const int64_t ckend = cw.kend;
        const int64_t k_stage = cw.k0;
        if (PROBE & 16) __builtin_amdgcn_s_setprio(1);       // experiment: MFMA section at raised wave priority
#pragma unroll
        for (int kk = 0; kk < BK; kk += 16) {
            if (!(PROBE & 4) && kk > 0 && k_stage + kk >= ckend) break;      // K tail: nothing but zero padding left
            float af[Cfg::MR][4], bf[Cfg::NR][4];
#pragma unroll
            for (int i = 0; i < Cfg::MR; ++i) {
                const int mrow = wm * Cfg::kWaveM + i * 16 + li;
                if constexpr (AT) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) af[i][t] = As[(kk + 4 * lg + t) * KStrided<BM, NTH>::kPitch + mrow];
                } else {
                    const float4 v = *reinterpret_cast<const float4*>(As + mrow * KPITCH + kk + 4 * lg);
                    af[i][0] = v.x; af[i][1] = v.y; af[i][2] = v.z; af[i][3] = v.w;
                }
            }
#pragma unroll
            for (int jn = 0; jn < Cfg::NR; ++jn) {
                const int ncol = wn * Cfg::kWaveN + jn * 16 + li;
                if constexpr (BT) {
                    const float4 v = *reinterpret_cast<const float4*>(Bs + ncol * KPITCH + kk + 4 * lg);
                    bf[jn][0] = v.x; bf[jn][1] = v.y; bf[jn][2] = v.z; bf[jn][3] = v.w;
                } else {
#pragma unroll
                    for (int t = 0; t < 4; ++t) bf[jn][t] = Bs[(kk + 4 * lg + t) * KStrided<BN, NTH>::kPitch + ncol];
                }
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < Cfg::MR; ++i)
#pragma unroll
                    for (int jn = 0; jn < Cfg::NR; ++jn)
                        // operands swapped (the 16x16 product is computed transposed): a lane then owns 4
                        // CONSECUTIVE COLUMNS of one row of C, and the epilogue stores float4s
                        acc[i][jn] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[jn][t], af[i][t], acc[i][jn], 0, 0, 0);
        }
        if (PROBE & 16) __builtin_amdgcn_s_setprio(0);
        // stage s+1: registers -> the other LDS buffer.  BEFORE the epilogue's stores in program order, so
        // that the wait for these registers never has to drain the C stores behind them.
        if (!(PROBE & 8)) sstore(cur ^ 1, sa, sb);
        if (ckt == ct.nk - 1) {
            // epilogue of this tile.  With the swapped operands the accumulator of lane (li, lg) holds
            // C[row = li][col = 4*lg + r], r = 0..3, of each 16x16 sub-tile: one float4 per sub-tile.
            // Columns in [N, roundup4(N)) are pad columns and are written as zeros (geogcn.h convention).
            float* Cout = ct.seg ? a.C[1] : a.C[0];
            int64_t ldc = ct.seg ? a.ldc[1] : a.ldc[0];
            const float* bias = ct.seg ? a.bias[1] : a.bias[0];
            const int64_t Nseg = ct.seg ? a.N[1] : a.N[0];
            const bool act_on = (ct.seg ? a.act_on[1] : a.act_on[0]) != 0;
            if constexpr (MODE == 1) {
                Cout = a.C[0] + (int64_t)ct.z * a.M * a.ldc[0] + (ct.seg ? a.slab_seg_w : 0);
                ldc = a.ldc[0];
            }
            const int64_t m0 = (int64_t)ct.mt * BM, n0 = (int64_t)ct.nt * BN;
            float bcol[Cfg::NR][4];
#pragma unroll
            for (int jn = 0; jn < Cfg::NR; ++jn) {
                const int64_t col0 = n0 + wn * Cfg::kWaveN + jn * 16 + lg * 4;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    bcol[jn][r] = 0.f;
                    if constexpr (MODE == 0) {
                        if (bias && col0 + r < Nseg) bcol[jn][r] = bias[col0 + r];
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < Cfg::MR; ++i) {
                const int64_t row = m0 + wm * Cfg::kWaveM + i * 16 + li;
                float* crow = Cout + row * ldc;
                const bool row_ok = row < a.M;
                // all loads of the epilogue (bias above, old C when accumulating) are issued as independent
                // batches before their first use -- no load/wait/store chains
                float4 oldv[Cfg::NR];
                if constexpr (MODE == 0) {
                    if (a.accumulate) {
#pragma unroll
                        for (int jn = 0; jn < Cfg::NR; ++jn) {
                            const int64_t col0 = n0 + wn * Cfg::kWaveN + jn * 16 + lg * 4;
                            oldv[jn] = make_float4(0.f, 0.f, 0.f, 0.f);
                            if (row_ok && col0 < Nseg) oldv[jn] = *reinterpret_cast<const float4*>(crow + col0);
                        }
                    }
                }
#pragma unroll
                for (int jn = 0; jn < Cfg::NR; ++jn) {
                    const int64_t col0 = n0 + wn * Cfg::kWaveN + jn * 16 + lg * 4;
                    float x[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        x[r] = acc[i][jn][r];
                        if constexpr (MODE == 0) {
                            x[r] += bcol[jn][r];
                            if (ACT == GEOGCN_ACT_NONE || act_on) x[r] = apply_act<ACT>(x[r]);
                        }
                    }
                    if constexpr (MODE == 0) {
                        if (a.accumulate) { x[0] += oldv[jn].x; x[1] += oldv[jn].y; x[2] += oldv[jn].z; x[3] += oldv[jn].w; }
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (col0 + r >= Nseg) x[r] = 0.f;
                    if ((PROBE & 2) ? (x[0] == 123.456f) : (row_ok && col0 < Nseg)) {
                        float* dst = crow + col0;
                        if (MODE == 0 && a.panel_w) {
                            const int64_t q = col0 / a.panel_w;
                            dst = Cout + (q * a.panel_R + row) * a.panel_w + (col0 - q * a.panel_w);
                        }
                        *reinterpret_cast<float4*>(dst) = make_float4(x[0], x[1], x[2], x[3]);
                    }
                    acc[i][jn] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
        }
        __syncthreads();
        cur ^= 1;
        if (++ckt == ct.nk) {
            ct = decode_tile(a, p, G, ++cj);
            ckt = 0;
        }
    };
    while (true) {
        step(ra0, rb0, ra1, rb1);            // set 1 holds s+1, set 0 is free for s+2
        if (ct.nk == 0) break;
        step(ra1, rb1, ra0, rb0);
        if (ct.nk == 0) break;
    }
}

// ---- A^T . B for the weight gradients WITHOUT LDS and without barriers (round 4) ------------------------------------------------
// dW = H^T . dZ reduces over the node dimension; both operands are k-STRIDED in memory ([K][M], [K][N]) -- and that is exactly the
// operand layout of v_mfma_f32_16x16x4_f32: lane l of an A (B) fragment holds H[k0 + l / 16][m0 + l % 16] (dZ[k0 + l / 16][n0 + l % 16]).
// So a fragment is ONE buffer_load_dword (four 64-byte row segments per wave), every wave streams its own MR + NR fragments per 4 k
// straight into registers, TN_DEPTH steps ahead, and multiplies MR x NR MFMAs -- no LDS image, no stage barrier (gemm_kernel<.., AT> spends
// a barrier, 61 KB of LDS stores and 40 ds_read_b32 per wave on every 32 k).  The 8 waves of a block (2 x 4: 160 x 320 / 128 x 256 ...)
// share lines through the CU's L1 only; the blocks of one K slab share an XCD (L2).  Split-K slabs go to the workspace in the layout
// splitk_reduce_kernel combines in slab order: deterministic.  tools/micro/tn_direct.hip: H^T . [dZ | dU] at the TwitterUS shape
// 1.36 ms = 116 TF against 1.67 ms = 94.5 TF for the staged kernel.  (k grouping per MFMA differs from the staged kernel's: results
// agree to fp32 rounding, not bit for bit.)
struct TnDirectArgs {
    int64_t M, K;
    const float* A; int64_t lda;
    const float* B[2]; int64_t ldb[2]; int64_t N[2];
    float* W; int64_t ldw, seg_w;       // slabs [nsplit][M][ldw]; N segment q starts at column q * seg_w
    int n_mt, n_nt, nt_per_seg, nsplit;
    int64_t kchunk;
};
constexpr int TN_DEPTH = 5;
// bytes one buffer descriptor bounds; the test seam GEOGCN_TN_SLAB_LIMIT (common.h) lowers it so that small operands reach the fallback
inline int64_t tn_slab_byte_limit() { return test_seam_i64("GEOGCN_TN_SLAB_LIMIT", 0x7FFFFFFFll); }

template <int MR, int NR>
__global__ __launch_bounds__(512, 1) void gemm_tn_direct_kernel(const TnDirectArgs a) {
    constexpr int WM = 2, WN = 4, D = TN_DEPTH;
    const int lane = threadIdx.x & 63;
    const int wid = uni(threadIdx.x >> 6);
    const int wm = wid / WN, wn = wid % WN;
    const int li = lane & 15, lk = lane >> 4;
    // block -> (tile, slab): the tiles of one K slab run on ONE XCD (block b runs on XCD b % 8), so that a slab's rows of A and B
    // come from HBM once and from that XCD's L2 for the other tiles
    const int T = a.n_mt * a.n_nt;
    const int b = blockIdx.x;
    const int zx = b % kNumXCD, rest = b / kNumXCD;
    const int tile = rest % T, z = (rest / T) * kNumXCD + zx;
    if (z >= a.nsplit) return;
    const int mt = tile / a.n_nt, ntile = tile % a.n_nt;
    const int seg = uni(ntile / a.nt_per_seg), nt = ntile % a.nt_per_seg;
    const int m0 = (mt * WM + wm) * MR * 16, n0 = (nt * WN + wn) * NR * 16;
    const int64_t M = a.M, N = seg ? a.N[1] : a.N[0];
    const float* Bp = seg ? a.B[1] : a.B[0];
    const int64_t ldb = seg ? a.ldb[1] : a.ldb[0];
    const int64_t kbeg = (int64_t)z * a.kchunk, kend = min(a.K, kbeg + a.kchunk);
    if (kbeg >= kend) return;
    // descriptors start at the slab's first row and end with its last: rows past the end read as zeros in hardware
    const __amdgpu_buffer_rsrc_t ar = tile_rsrc(a.A + kbeg * a.lda, (kend - kbeg) * a.lda * 4);
    const __amdgpu_buffer_rsrc_t br = tile_rsrc(Bp + kbeg * ldb, (kend - kbeg) * ldb * 4);
    // lane part of the offsets; a lane whose column lies beyond the matrix reads zeros (its pitch may have no pad columns)
    uint32_t ao[MR], bo[NR];
#pragma unroll
    for (int i = 0; i < MR; ++i) ao[i] = (m0 + i * 16 + li < M) ? (uint32_t)((lk * a.lda + m0 + i * 16 + li) * 4) : kOobOffset;
#pragma unroll
    for (int j = 0; j < NR; ++j) bo[j] = (n0 + j * 16 + li < N) ? (uint32_t)((lk * ldb + n0 + j * 16 + li) * 4) : kOobOffset;
    const uint32_t astep = (uint32_t)a.lda * 16u, bstep = (uint32_t)ldb * 16u;          // 4 rows per step, in bytes
    f32x4 acc[MR][NR];
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nsteps = (int)((kend - kbeg + 3) / 4);
    float ra[D + 1][MR], rb[D + 1][NR];
    auto fetch = [&](float (&fa)[MR], float (&fb)[NR], int s) {          // (steps past the end: beyond num_records -> zeros)
#pragma unroll
        for (int i = 0; i < MR; ++i)
            fa[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ar, (int)ao[i], (int)((uint32_t)s * astep), 0));
#pragma unroll
        for (int j = 0; j < NR; ++j)
            fb[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(br, (int)bo[j], (int)((uint32_t)s * bstep), 0));
    };
#pragma unroll
    for (int d = 0; d < D; ++d) fetch(ra[d], rb[d], d);
#pragma unroll 1
    for (int s0 = 0; s0 < nsteps; s0 += D + 1) {
#pragma unroll
        for (int u = 0; u <= D; ++u) {
            fetch(ra[(u + D) % (D + 1)], rb[(u + D) % (D + 1)], s0 + u + D);
            if (s0 + u < nsteps) {
#pragma unroll
                for (int i = 0; i < MR; ++i)
#pragma unroll
                    for (int j = 0; j < NR; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ra[u][i], rb[u][j], acc[i][j], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // D[i = 4 * (lane / 16) + r][j = lane % 16]  ->  slab z, row m, column n of segment seg
    float* Wz = a.W + (int64_t)z * M * a.ldw + (seg ? a.seg_w : 0);
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t m = m0 + i * 16 + 4 * lk + r, n = n0 + j * 16 + li;
                if (m < M && n < N) Wz[m * a.ldw + n] = acc[i][j][r];
            }
}

// ---- the highway block's fused products on WHOLE ROWS of A (round 3) ---------------------------------------------------
// For N_nodes x 300 x [300 | 300] (the dual launch) and dZ.Wh^T + dU.Wt^T (the k-concatenated one) the operand that matters is
// skinny: K = 300.  A block takes 64 whole rows of A -- one contiguous 77 KB read into LDS (64 x 308 floats: two blocks per CU) --
// and no barrier follows until the tile is done: every wave multiplies all 64 rows by its own 80 columns per pass (5 column
// tiles, 80 accumulator registers), B fragments coming straight from the L2-resident weights, which a prep kernel lays out in
// FRAGMENT order ([column tile][16-k step][lane][4 floats]: a wave's fragment load is 1 KB of consecutive bytes), two steps
// ahead.  The instruction stream of a step is 80 MFMAs, 4 ds_read_b128 and 5 buffer loads -- the staged kernel above spends a
// barrier, two LDS images and 18 fragment reads on 160.  Same MFMA and the same k grouping as gemm_kernel (lane (li, lg) holds
// k = 16 s + 4 lg + t for the t-th MFMA of step s), one accumulator over both reductions of the k-concatenated form: results
// are bit-identical.  tools/micro/f32_astat.hip: the dual launch 1.41 ms = 112 TF against 1.65 ms = 96 TF.
__global__ __launch_bounds__(TPB) void prep_b_frag_f32_kernel(const float* __restrict__ W, int64_t ldw, int K, int N, int NK,
                                                              int n_tiles, int b_is_nk, float* __restrict__ out) {
    const int64_t total = (int64_t)n_tiles * NK * 256;
    for (int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x; e < total; e += (int64_t)gridDim.x * TPB) {
        const int el = (int)(e & 3), lane = (int)((e >> 2) & 63);
        const int64_t f = e >> 8;
        const int kt = (int)(f % NK), nt = (int)(f / NK);
        const int n = nt * 16 + (lane & 15), k = kt * 16 + (lane >> 4) * 4 + el;
        out[e] = (k < K && n < N) ? (b_is_nk ? W[(int64_t)n * ldw + k] : W[(int64_t)k * ldw + n]) : 0.f;
    }
}

constexpr int kRowsBM = 64, kRowsWCT = 5, kRowsDepth = 2;

struct RowsArgs {
    int64_t M;
    int n_mt;
    int n_kseg;                         // 1, or 2 A operands reduced into one accumulator (then ONE column pass)
    const float* A[2]; int64_t lda[2]; int K[2];
    const float* Bf[2];                 // fragment-ordered weights of slot q = N segment | K segment
    int n_nseg; int passes[2];          // column passes of each N segment (4 waves x wct tiles x 16 columns per pass)
    int wct[2];                         // column tiles a wave takes per pass: 5, or 4 where 4 x passes x 4 tiles cover the segment (N = 256, 512)
    float* C[2]; int64_t ldc[2]; const float* bias[2]; int64_t N[2]; int act_on[2];
    int accumulate;
    // C += G * (1 - T), each operation rounded on its own: the carry gradient of the highway block formed here instead of being
    // written by highway_bwd and read back (NULL: none)
    const float* gateG; int64_t ldg; const float* gateT; int64_t ldt;
    // ... and, on top of it, C = that * (keep * scale) * (1 - Y^2): the dropout + tanh gradient of the layer below the first highway
    // block (what geogcn_act_bwd_f32 would do in a pass of its own); keep = bytes, pitch postF (a multiple of 4)
    const float* postY; int64_t ldy; const uint8_t* postKeep; int64_t postF; float postScale;
};

template <int KP, int ACT, bool GATE = false, bool POST = false>
__global__ __launch_bounds__(TPB, 2) void gemm_rows_kernel(const RowsArgs a) {
    constexpr int BM = kRowsBM, WCT = kRowsWCT, DEPTH = kRowsDepth, D1 = DEPTH + 1;
    constexpr int PITCH = KP + 4;               // floats per LDS row: an odd number of float4s -> conflict-free ds_read_b128
    constexpr int F4R = KP / 4;
    constexpr int ITERS = BM * F4R / TPB;
    constexpr int MR = BM / 16, NK = KP / 16;
    static_assert(BM * F4R % TPB == 0, "a tile must divide over the block");
    extern __shared__ __attribute__((aligned(16))) float As[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int P = a.n_nseg == 2 ? a.passes[0] + a.passes[1] : a.passes[0];
    for (int mt = blockIdx.x; mt < a.n_mt; mt += gridDim.x) {
        const int64_t m0 = (int64_t)mt * BM;
        f32x4 acc[MR][WCT];
#pragma unroll 1
        for (int ks = 0; ks < a.n_kseg; ++ks) {
            if (ks) __syncthreads();            // everybody done with the first operand's rows
            {
                // (the offsets are the same for every tile: left alone, hipcc computes them once and keeps ~40 registers alive
                //  across the MFMA loop -- an opaque copy of the thread index makes them per-tile work)
                int tt = tid;
                asm volatile("" : "+v"(tt));
                const float* Ap = ks ? a.A[1] : a.A[0];
                const int64_t lda = ks ? a.lda[1] : a.lda[0];
                const int K4 = ((ks ? a.K[1] : a.K[0]) + 3) & ~3;       // pad columns up to roundup4(K) are zero; beyond: not read
                const __amdgpu_buffer_rsrc_t rs = tile_rsrc(Ap + m0 * lda, std::min<int64_t>(BM, a.M - m0) * lda * 4);
                const uint32_t ld4 = (uint32_t)lda * 4u;
                float4 v[ITERS];
#pragma unroll
                for (int i = 0; i < ITERS; ++i) {
                    const int idx = tt + TPB * i;
                    const int r = idx / F4R, c = idx - r * F4R;
                    v[i] = buffer_load4(rs, c * 4 < K4 ? (uint32_t)r * ld4 + (uint32_t)c * 16u : kOobOffset);
                }
#pragma unroll
                for (int i = 0; i < ITERS; ++i) {
                    const int idx = tt + TPB * i;
                    const int r = idx / F4R, c = idx - r * F4R;
                    *reinterpret_cast<float4*>(As + r * PITCH + c * 4) = v[i];
                }
            }
            __syncthreads();
#pragma unroll 1
            for (int ps = 0; ps < P; ++ps) {
                const int seg = ps >= a.passes[0] ? 1 : 0;                  // wave-uniform
                const int lps = seg ? ps - a.passes[0] : ps;
                // column group of this wave: ROTATED with the row tile, so that the group whose last column tile is all padding
                // (300 columns = 19 tiles of 16 over 4 waves: 5, 5, 5, 4) visits every SIMD in turn -- the two co-resident
                // blocks of a CU then share MFMA pipes that carry 4.75 instead of 5 tiles per wave on average
                const int cg = (wid + mt) & 3;
                const int w = seg ? a.wct[1] : a.wct[0];
                const int tile0 = (cg * (seg ? a.passes[1] : a.passes[0]) + lps) * w;
                // does the LAST of this wave's column tiles hold any real column OF ITS OWN?  (wave-uniform: a scalar branch per k-step)
                const bool last_real = __builtin_amdgcn_readfirstlane((int)(w == WCT && (int64_t)(tile0 + WCT - 1) * 16 < (seg ? a.N[1] : a.N[0]))) != 0;
                const int slot = seg | ks;
                const int n_tiles = 4 * (seg ? a.passes[1] : a.passes[0]) * WCT;
                const __amdgpu_buffer_rsrc_t brs = tile_rsrc(slot ? a.Bf[1] : a.Bf[0], (int64_t)n_tiles * NK * 1024);
                if (ks == 0) {
#pragma unroll
                    for (int i = 0; i < MR; ++i)
#pragma unroll
                        for (int j = 0; j < WCT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
                // B fragments DEPTH steps ahead in a ring of DEPTH + 1 register sets; the k loop stays ROLLED (DEPTH + 1 steps
                // per trip, ring slots are compile-time constants inside a trip) and the scheduler is fenced per step; requests
                // past the last step re-read the last one (no branch around a load)
                auto bload = [&](f32x4 (&b)[WCT], int kt) {
                    const int kk = kt < NK ? kt : NK - 1;
#pragma unroll
                    for (int j = 0; j < WCT; ++j)
                        b[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(brs, lane * 16, ((tile0 + j) * NK + kk) * 1024, 0));
                };
                auto kstep = [&](const f32x4 (&b)[WCT], int kt) {
                    f32x4 af[MR];
#pragma unroll
                    for (int i = 0; i < MR; ++i) af[i] = *reinterpret_cast<const f32x4*>(As + (i * 16 + li) * PITCH + kt * 16 + lg * 4);
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int i = 0; i < MR; ++i)
#pragma unroll
                            for (int j = 0; j < WCT - 1; ++j)
                                // operands swapped (as in gemm_kernel): a lane owns 4 consecutive columns of one row of C
                                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[j][t], af[i][t], acc[i][j], 0, 0, 0);
                    if (last_real) {          // (an all-padding tile would only multiply zeros: its accumulators stay 0, its stores are masked)
#pragma unroll
                        for (int t = 0; t < 4; ++t)
#pragma unroll
                            for (int i = 0; i < MR; ++i)
                                acc[i][WCT - 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[WCT - 1][t], af[i][t], acc[i][WCT - 1], 0, 0, 0);
                    }
                };
                f32x4 ring[D1][WCT];
#pragma unroll
                for (int d = 0; d < DEPTH; ++d) bload(ring[d], d);
#pragma unroll 1
                for (int k0 = 0; k0 < NK; k0 += D1) {
#pragma unroll
                    for (int u = 0; u < D1; ++u) {
                        bload(ring[(u + DEPTH) % D1], k0 + u + DEPTH);
                        if (k0 + u < NK) kstep(ring[u], k0 + u);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                if (ks != a.n_kseg - 1) continue;
                // epilogue: the arithmetic of gemm_kernel, in its order
                float* Cout = seg ? a.C[1] : a.C[0];
                const int64_t ldc = seg ? a.ldc[1] : a.ldc[0];
                const float* bias = seg ? a.bias[1] : a.bias[0];
                // (columns from tile0 + w on belong to the next wave)
                const int64_t Nseg = std::min<int64_t>(seg ? a.N[1] : a.N[0], (int64_t)(tile0 + w) * 16);
                const bool act_on = (seg ? a.act_on[1] : a.act_on[0]) != 0;
                const int64_t ncol0 = (int64_t)tile0 * 16;
                float bcol[WCT][4];
#pragma unroll
                for (int j = 0; j < WCT; ++j) {
                    const int64_t col0 = ncol0 + j * 16 + lg * 4;
#pragma unroll
                    for (int r = 0; r < 4; ++r) bcol[j][r] = (bias && col0 + r < Nseg) ? bias[col0 + r] : 0.f;
                }
                // what the epilogue reads besides the accumulators -- the old C (accumulate) or G and T (gate carry) -- is requested
                // one row tile AHEAD of its use, column tile by column tile as the registers of the current one fall free (the
                // compiler cannot move a load above the stores before it: C may alias anything)
                // (locals, not fields of the by-value argument struct: a lambda that touches `a` makes the compiler keep a copy of it in scratch)
                const float* const gG = GATE ? a.gateG : nullptr; const float* const gT = GATE ? a.gateT : nullptr;      // (GATE: its own instance -- the second operand array costs 20 registers)
                const int64_t ldg = a.ldg, ldt = a.ldt, Mrows = a.M;
                const bool accum = a.accumulate != 0;
                const bool extra = accum || gG != nullptr;
                f32x4 ea[WCT], eb[WCT];
                // (buffer loads: one descriptor per operand for the row tile, a 32-bit offset per request -- 64-bit addresses for
                //  every (row tile, column tile) pair had the kernel at 256 registers with spills; rows past M read as zero)
                const int64_t lda_e = gG ? ldg : ldc;
                const int64_t rows_here = std::min<int64_t>(BM, Mrows - m0);
                const __amdgpu_buffer_rsrc_t ersA = tile_rsrc((gG ? gG + m0 * ldg : Cout + m0 * ldc), extra ? rows_here * lda_e * 4 : 0);
                const __amdgpu_buffer_rsrc_t ersB = tile_rsrc(gG ? gT + m0 * ldt : Cout, gG ? rows_here * ldt * 4 : 0);
                const uint32_t lda_e4 = (uint32_t)lda_e * 4u, ldt4 = (uint32_t)ldt * 4u;
                const __amdgpu_buffer_rsrc_t prsY = tile_rsrc(POST ? a.postY + m0 * a.ldy : Cout, POST ? rows_here * a.ldy * 4 : 0);
                const __amdgpu_buffer_rsrc_t prsK = tile_rsrc(POST ? reinterpret_cast<const float*>(a.postKeep + m0 * a.postF) : Cout,
                                                              POST ? rows_here * a.postF : 0);
                const uint32_t ldy4 = POST ? (uint32_t)a.ldy * 4u : 0u, pF = POST ? (uint32_t)a.postF : 0u;
                const float pscale = POST ? a.postScale : 0.f;
                auto eload = [&](int i, int j) __attribute__((always_inline)) {
                    const int64_t col0 = ncol0 + j * 16 + lg * 4;
                    const uint32_t r = (uint32_t)(i * 16 + li), c = (uint32_t)col0 * 4u;
                    const bool ok = col0 < Nseg;
                    ea[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ersA, (int)(ok ? r * lda_e4 + c : kOobOffset), 0, 0));
                    // (POST: T is requested where it is used -- twenty registers the third flavour does not have)
                    if constexpr (GATE && !POST) eb[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ersB, (int)(ok ? r * ldt4 + c : kOobOffset), 0, 0));
                };
                if (extra) {
#pragma unroll
                    for (int j = 0; j < WCT; ++j) eload(0, j);
                }
                // (POST: T, Y and the keep bytes one column tile ahead -- nine registers; rows past M / columns past N read as zero,
                //  their results are masked before the store anyway)
                f32x4 pt = f32x4{0.f, 0.f, 0.f, 0.f}, py = pt;
                uint32_t pk = 0;
                auto pload = [&](int i, int j) __attribute__((always_inline)) {
                    const int64_t col0 = ncol0 + j * 16 + lg * 4;
                    const uint32_t pr = (uint32_t)(i * 16 + li), pc = (uint32_t)col0;
                    const bool pok = col0 < Nseg;
                    pt = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ersB, (int)(pok ? pr * ldt4 + pc * 4u : kOobOffset), 0, 0));
                    py = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(prsY, (int)(pok ? pr * ldy4 + pc * 4u : kOobOffset), 0, 0));
                    pk = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(prsK, (int)(pok ? pr * pF + pc : kOobOffset), 0, 0);
                };
                if constexpr (POST) pload(0, 0);
#pragma unroll
                for (int i = 0; i < MR; ++i) {
                    const int64_t row = m0 + i * 16 + li;
                    float* crow = Cout + row * ldc;
                    const bool row_ok = row < Mrows;
#pragma unroll
                    for (int j = 0; j < WCT; ++j) {
                        const int64_t col0 = ncol0 + j * 16 + lg * 4;
                        float x[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            x[r] = acc[i][j][r] + bcol[j][r];
                            if (ACT == GEOGCN_ACT_NONE || act_on) x[r] = apply_act<ACT>(x[r]);
                        }
                        if (accum) { x[0] += ea[j][0]; x[1] += ea[j][1]; x[2] += ea[j][2]; x[3] += ea[j][3]; }
                        if constexpr (GATE) {
                            f32x4 tq;
                            if constexpr (POST) tq = pt;
                            else tq = eb[j];
                            x[0] = add_gate_carry(x[0], ea[j][0], tq[0]); x[1] = add_gate_carry(x[1], ea[j][1], tq[1]);
                            x[2] = add_gate_carry(x[2], ea[j][2], tq[2]); x[3] = add_gate_carry(x[3], ea[j][3], tq[3]);
                        }
                        if constexpr (POST) {
                            const f32x4 yv = py;
                            const uint32_t kv = pk;
                            x[0] = masked_tanh_bwd(x[0], (float)(kv & 0xffu), pscale, yv[0]);
                            x[1] = masked_tanh_bwd(x[1], (float)((kv >> 8) & 0xffu), pscale, yv[1]);
                            x[2] = masked_tanh_bwd(x[2], (float)((kv >> 16) & 0xffu), pscale, yv[2]);
                            x[3] = masked_tanh_bwd(x[3], (float)(kv >> 24), pscale, yv[3]);
                        }
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (col0 + r >= Nseg) x[r] = 0.f;
                        if (row_ok && col0 < Nseg) *reinterpret_cast<float4*>(crow + col0) = make_float4(x[0], x[1], x[2], x[3]);
                        // this column tile's operands of the NEXT row tile: on their way while the remaining tiles are finished
                        if (extra && i + 1 < MR) eload(i + 1, j);
                        if constexpr (POST) {           // ... and the third flavour's three operands of the NEXT column tile
                            if (j + 1 < WCT) pload(i, j + 1);
                            else if (i + 1 < MR) pload(i + 1, 0);
                        }
                    }
                }
            }
        }
        __syncthreads();          // everybody done reading this tile's rows
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
    // slabs added in slab order; eight loads are in flight at a time (a load -> add chain per slab paid one L2 round trip each)
    float acc = 0.f;
    const float* w = W + row * ldw + col;
    const int64_t zs = M * ldw;
    int z = 0;
    for (; z + 8 <= nsplit; z += 8) {
        float pv[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) pv[k] = w[(int64_t)(z + k) * zs];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += pv[k];
    }
    for (; z < nsplit; ++z) acc += w[(int64_t)z * zs];
    if (bias) acc += bias[col];
    acc = apply_act<ACT>(acc);
    if (accumulate) acc += C[row * ldc + col];
    C[row * ldc + col] = acc;
}

}  // namespace

int splitk_reduce_launch(int64_t M, int64_t N, int nsplit, const float* W, int64_t ldw, float* C, int64_t ldc,
                         const float* bias, int act, int accumulate, hipStream_t st) {
    const dim3 rgrid((unsigned)cdiv(M * N, TPB));
#define GEOGCN_RED(ACT)                                                                                 \
    hipLaunchKernelGGL((splitk_reduce_kernel<ACT>), rgrid, dim3(TPB), 0, st, M, N, nsplit, W, ldw, C, ldc, bias, \
                       accumulate)
    if (act == GEOGCN_ACT_TANH) GEOGCN_RED(GEOGCN_ACT_TANH);
    else if (act == GEOGCN_ACT_SIGMOID) GEOGCN_RED(GEOGCN_ACT_SIGMOID);
    else GEOGCN_RED(GEOGCN_ACT_NONE);
#undef GEOGCN_RED
    GEOGCN_LAUNCH_CHECK("splitk_reduce_kernel");
    return 0;
}

namespace {

struct SplitPlan {
    int nsplit;
    int64_t kchunk;
    int grid;
};

// grid = resident persistent blocks; transA additionally slices K so that (tiles x slices) fills the grid
template <int BM, int BN, bool AT, bool BT, int WM = 2, int WN = 2>
SplitPlan plan_grid(int64_t M, int64_t n_nt, int64_t K) {
    const int64_t tiles = cdiv(M, BM) * n_nt;
    const int G = kNumCU * GemmCfg<BM, BN, AT, BT, WM, WN>::kBlocksPerCU;
    SplitPlan sp{1, cdiv(K, BK) * BK, 0};
    if (AT) {
        int64_t ns = std::max<int64_t>(1, G / tiles);
        // at least 4 stages (128 reduction rows) per slice: a short reduction (CMU shape: K = 9,475) is spread
        // over as many CUs as that allows instead of 18 slices of 16 stages on 256 CUs
        const int64_t max_ns = std::max<int64_t>(1, K / (BK * 4));
        ns = std::min(ns, max_ns);
        if constexpr (WM == 2 && WN == 4) {
            // (round 6) the wide tiles run one 8-wave block per CU with all tiles of a K slab on ONE XCD (32 CUs).  More tiles than that
            // -- 900 x [900 | 900]: 36 -- and one slab per XCD takes two rounds with the second almost empty (measured: 0.21 of the bf16
            // peak against 0.38 at 300 wide): take s slabs per XCD, s <= 8, so that tiles x s fills whole rounds of 32
            constexpr int64_t cus = kNumCU / kNumXCD;
            if (tiles > cus) {
                int64_t best_s = 1, best_num = 0, best_den = 1;
                for (int64_t sx = 1; sx <= 8; ++sx) {
                    if (sx * kNumXCD > max_ns) break;
                    const int64_t num = tiles * sx, den = cus * cdiv(tiles * sx, cus);          // efficiency num / den
                    if (num * best_den > best_num * den) { best_s = sx; best_num = num; best_den = den; }
                }
                ns = best_s * kNumXCD;
            }
        }
        sp.kchunk = cdiv(cdiv(K, ns), BK) * BK;
        sp.nsplit = (int)cdiv(K, sp.kchunk);
    }
    const int64_t total = tiles * sp.nsplit;
    sp.grid = (int)std::min<int64_t>(G, cdiv(total, kNumXCD) * kNumXCD);
    return sp;
}

template <int BM, int BN, bool AT, bool BT, int WM = 2, int WN = 2>
int launch_gemm(const GemmCall& c, void* ws, size_t ws_bytes, hipStream_t st) {
    using Cfg = GemmCfg<BM, BN, AT, BT, WM, WN>;
    const int nt_per_seg = (int)cdiv(c.maxN(), BN);
    const int n_nt = nt_per_seg * c.n_nseg;
    const SplitPlan sp = plan_grid<BM, BN, AT, BT, WM, WN>(c.M, n_nt, c.K[0]);
    GemmArgs a{};
    a.M = c.M;
    for (int q = 0; q < 2; ++q) {
        a.A[q] = c.A[q]; a.lda[q] = c.lda[q]; a.B[q] = c.B[q]; a.ldb[q] = c.ldb[q]; a.C[q] = c.C[q]; a.ldc[q] = c.ldc[q];
        a.bias[q] = c.bias[q]; a.N[q] = c.N[q]; a.K[q] = c.K[q]; a.act_on[q] = c.act[q] != GEOGCN_ACT_NONE;
    }
    a.accumulate = c.accumulate;
    a.panel_w = c.panel_w;
    a.panel_R = c.panel_R;
    a.kchunk = sp.kchunk;
    a.n_mt = (int)cdiv(c.M, BM);
    a.n_nt = n_nt;
    a.nt_per_seg = nt_per_seg;
    a.n_split = sp.nsplit;
    a.n_kseg = c.n_kseg;
    a.nk0 = (int)cdiv(c.K[0], BK);
    a.xcd_order = (a.n_mt >= 4 * kNumXCD) ? 1 : 0;
    const int act = c.act[0] != GEOGCN_ACT_NONE ? c.act[0] : c.act[1];     // (the entry points check that they agree)
    const dim3 grid((unsigned)sp.grid);
#define GEOGCN_GEMM_LAUNCH(ACT, MODE)                                                                    \
    do {                                                                                                  \
        auto kern = gemm_kernel<BM, BN, AT, BT, ACT, MODE, 0, WM, WN>;                                    \
        static LdsAttrOnce lds_once;                                                                    \
        if (const int rc_ = lds_once.ensure((const void*)kern, (int)Cfg::kLdsBytes)) return rc_; \
        hipLaunchKernelGGL(kern, grid, dim3(Cfg::NTH), Cfg::kLdsBytes, st, a);                             \
        GEOGCN_LAUNCH_CHECK("gemm_kernel");                                                               \
    } while (0)

    if (sp.nsplit == 1) {
        if (act == GEOGCN_ACT_TANH) GEOGCN_GEMM_LAUNCH(GEOGCN_ACT_TANH, 0);
        else if (act == GEOGCN_ACT_SIGMOID) GEOGCN_GEMM_LAUNCH(GEOGCN_ACT_SIGMOID, 0);
        else GEOGCN_GEMM_LAUNCH(GEOGCN_ACT_NONE, 0);
        return 0;
    }
    // split-K slabs into the workspace, then the ordered combine (per N segment)
    GEOGCN_REQUIRE(c.n_kseg == 1, GEOGCN_E_ARG, "gemm_f32: split-K with two K segments");
    const int64_t seg_w = (c.maxN() + 3) & ~(int64_t)3;      // slabs are stored as float4s
    const int64_t ldw = seg_w * c.n_nseg;
    const size_t need = (size_t)sp.nsplit * (size_t)c.M * (size_t)ldw * sizeof(float);
    GEOGCN_REQUIRE(ws && ws_bytes >= need, GEOGCN_E_ARG, "gemm_f32: split-K workspace too small (%zu < %zu)",
                   ws_bytes, need);
    float* W = (float*)ws;
    int nsplit_used = sp.nsplit;
    bool direct_done = false;
    if constexpr (AT && !BT && WM == 2 && WN == 4) {
        // one 8-wave block per CU and the tiles of a slab on one XCD: an XCD (32 CUs) takes ceil(nsplit / 8) * T blocks -- keep that
        // within 32, or the 33rd block of an XCD runs after the others (384 x 300: 3 tiles x 85 slabs took twice the time)
        const int T = a.n_mt * n_nt;
        int ns = sp.nsplit;
        const int ns_cap = kNumXCD * std::max(1, (kNumCU / kNumXCD) / T);
        if (T <= kNumCU / kNumXCD && ns > ns_cap) ns = ns_cap;          // (more tiles than an XCD has CUs: plan_grid chose whole rounds)
        const int64_t kchunk = ns == sp.nsplit ? sp.kchunk : cdiv(cdiv(c.K[0], ns), BK) * BK;
        const int nsplit = (int)cdiv(c.K[0], kchunk);
        // (a slab's rows must lie within one 2 GB buffer descriptor for its end to be a hardware bound -- checked on the slab these kernels
        //  actually take, after the cap above may have enlarged it; wider operands than the GCN's stay on the staged kernel)
        const int64_t max_ld = std::max(c.lda[0], std::max(c.ldb[0], c.n_nseg == 2 ? c.ldb[1] : 0));
        const bool slab_ok = kchunk * max_ld * 4 < tn_slab_byte_limit();
        if (slab_ok && c.precision == GEOGCN_GEMM_BF16X3 && x3_tn_takes(BM, BN)) {
            // fp32-class split-bf16 products, operands transposed + split on their way into LDS (gemm_x3.hip)
            X3TnCall t{};
            t.M = c.M; t.K = c.K[0]; t.A = c.A[0]; t.lda = c.lda[0];
            for (int q = 0; q < 2; ++q) { t.B[q] = c.B[q]; t.ldb[q] = c.ldb[q]; t.N[q] = c.N[q]; }
            t.W = W; t.ldw = ldw; t.seg_w = seg_w;
            t.n_mt = a.n_mt; t.n_nt = n_nt; t.nt_per_seg = nt_per_seg;
            t.kchunk = kchunk; t.nsplit = nsplit;
            if (const int rc = x3_tn_launch(BM, BN, t, st)) return rc;
            nsplit_used = nsplit;
            direct_done = true;
        }
#ifndef GEOGCN_F32_NO_TN_DIRECT          // (A/B build only: every exact A^T . B on the staged kernel)
        if (slab_ok && !direct_done) {
            // the weight gradients: fragments straight from L1 / L2 into registers, no LDS, no barriers (gemm_tn_direct_kernel)
            TnDirectArgs t{};
            t.M = c.M; t.K = c.K[0]; t.A = c.A[0]; t.lda = c.lda[0];
            for (int q = 0; q < 2; ++q) { t.B[q] = c.B[q]; t.ldb[q] = c.ldb[q]; t.N[q] = c.N[q]; }
            t.W = W; t.ldw = ldw; t.seg_w = seg_w;
            t.n_mt = a.n_mt; t.n_nt = n_nt; t.nt_per_seg = nt_per_seg;
            t.kchunk = kchunk; t.nsplit = nsplit;
            nsplit_used = nsplit;
            const dim3 dgrid((unsigned)(cdiv(t.nsplit, kNumXCD) * kNumXCD * T));
            hipLaunchKernelGGL((gemm_tn_direct_kernel<BM / 32, BN / 64>), dgrid, dim3(512), 0, st, t);
            GEOGCN_LAUNCH_CHECK("gemm_tn_direct_kernel");
            direct_done = true;
        }
#endif
    }
    if (!direct_done) {
    a.C[0] = W;
    a.ldc[0] = ldw;
    a.slab_seg_w = seg_w;
    a.bias[0] = a.bias[1] = nullptr;
    a.accumulate = 0;
    GEOGCN_GEMM_LAUNCH(GEOGCN_ACT_NONE, 1);
    }
    for (int q = 0; q < c.n_nseg; ++q) {
        const int rc = splitk_reduce_launch(c.M, c.N[q], nsplit_used, W + q * seg_w, ldw, c.C[q], c.ldc[q], c.bias[q],
                                            c.act[q], c.accumulate, st);
        if (rc) return rc;
    }
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

// 8-wave "wide" tiles for transA (dW = H^T.dZ): one tile spans all of N (161..320 columns), 2 x 4 waves, one
// block per CU.  Both streamed operands are staged once instead of once per N tile, and the 160x160 tile of the
// 4-wave kernel only fits one block (= one wave per SIMD) per CU: 0.83 -> 0.755 ms at 440000x300x300.  For
// NN / NT the 4-wave tiles with two blocks per CU measured 3 % faster, so those keep them.
inline int wide_bn(int64_t N, int precision = GEOGCN_GEMM_F32) {
    if (N > 256 && N <= 320) return 320;
    if (N > 160 && N <= 256) return 256;
    // (round 6) wider weight gradients under bf16x3 -- the reference's WORLD run, README.md:177-181: -hid 900 900 900, 930 classes -- in
    // several wide tiles, so that gemm_x3.hip's A^T . B kernel takes them (900 -> 3 x 320, 512 -> 2 x 256); exact fp32 keeps its 4-wave tiles
    if (N > 320 && precision == GEOGCN_GEMM_BF16X3) return cdiv(N, 320) * 320 <= cdiv(N, 256) * 256 ? 320 : 256;
    return 0;
}

template <bool AT, bool BT>
int dispatch_tiles(int bm, int bn, const GemmCall& c, void* ws, size_t ws_bytes, hipStream_t st) {
#define GEOGCN_T(BM_, BN_) \
    if (bm == BM_ && bn == BN_) return launch_gemm<BM_, BN_, AT, BT>(c, ws, ws_bytes, st);
#define GEOGCN_W(BM_, BN_) \
    if (bm == BM_ && bn == BN_) return launch_gemm<BM_, BN_, AT, BT, 2, 4>(c, ws, ws_bytes, st);
    GEOGCN_T(128, 128)
    GEOGCN_T(128, 160)
    if constexpr (!AT) {
        GEOGCN_T(64, 128)           // few rows (CMU shape): twice the tiles, so that more than half the CUs get one
        GEOGCN_T(64, 160)
    }
    if constexpr (!AT) {
        GEOGCN_T(96, 160)           // A . B^T: two k-contiguous images of 128+160 rows would not fit twice in 160 KB;
                                    // A . B: mid-size operands (choose_tiles)
    }
    if constexpr (AT) {
        GEOGCN_T(160, 128)
        GEOGCN_T(160, 160)
        GEOGCN_W(128, 256)
        GEOGCN_W(128, 320)
        GEOGCN_W(160, 256)
        GEOGCN_W(160, 320)
    }
#undef GEOGCN_T
#undef GEOGCN_W
    set_error("gemm_f32: no kernel for tile %dx%d", bm, bn);
    return GEOGCN_E_ARG;
}

// tile shape of a call: (bm, bn)
inline void choose_tiles(bool transA, bool transB, int64_t M, int64_t maxN, int n_nseg, int& bm, int& bn, int precision = GEOGCN_GEMM_F32) {
    const int wbn = transA ? wide_bn(maxN, precision) : 0;
    bn = wbn ? wbn : pick_tile(maxN);
    if (transA) {
        bm = pick_tile(M);
        return;
    }
    // short operands: with 128-row tiles fewer tiles than CUs -> 64-row tiles (M = 9,475: 150 -> 298 tiles)
    const bool few = cdiv(M, 128) * cdiv(maxN, bn) * n_nseg < kNumCU;
    bm = few ? 64 : ((transB && bn == 160) ? 96 : 128);
    // in between (one to two rounds of the 512 resident blocks: the CMU shape's fused forward pair is 75 x 4 tiles of
    // 128 rows) a 96-row tile that fits ONE round beats both: rounds x rows per tile is what the launch takes
    if (!few && !transB && bn == 160) {
        const int64_t slots = 2 * kNumCU, nt = cdiv(maxN, bn) * n_nseg;
        const int64_t c128 = cdiv(cdiv(M, 128) * nt, slots) * 128, c96 = cdiv(cdiv(M, 96) * nt, slots) * 96;
        // (only there: over many rounds the larger tile's lower operand traffic wins -- 440,000 x 300 x 300: 0.81 ms with
        //  128 rows, 0.85 with 96)
        if (c96 < c128 && cdiv(M, 128) * nt <= 2 * slots) bm = 96;
    }
}

// ---- whole-rows kernel: which calls take it, its workspace, its launch ---------------------------------------------------------
// The fused launches (two N segments or two K segments) and, since round 4, single A . B^T products (rows_kp); a single 300-wide
// A . B runs as fast on the staged kernel (0.74 against 0.75 ms).  Every segment must fill its column passes (a pass is 320 columns: 300 -> 320 is the padding the
// staged kernel has as well; 256 would multiply 64 columns of zeros: at most an eighth may be padding) and K must pad to an instantiated depth.
#ifndef GEOGCN_ROWS_MIN_M
#define GEOGCN_ROWS_MIN_M 32768
#endif
constexpr int64_t kRowsMinM = GEOGCN_ROWS_MIN_M;          // rows from which the whole-rows kernel is taken (A/B builds override)
inline int rows_passes(int64_t N) { return N <= 4 * kRowsWCT * 16 ? 1 : 2; }
inline int rows_wct(int64_t N) { return (int)cdiv(cdiv(N, 16), 4 * rows_passes(N)); }      // 5, or 4 (N <= 256, 321..512), or fewer
inline int rows_kp(const GemmCall& c, bool transA, bool transB = false) {
#ifdef GEOGCN_F32_NO_ROWS_KERNEL          // A/B build only (GEOGCN_BUILD_DEFINES): every call on the staged kernel
    return 0;
#endif
    // (round 4) ... and the single A . B^T products (dH = dS . W^T of the output layer and of plain layers): the staged kernel's
    // weakest form (96 x 160 tiles, 0.57 of the MFMA peak at 440,000 x 300 x 256); here the weights are laid out in fragment
    // order whatever their orientation
    // ... and single A . B products whose width four waves cover with FOUR column tiles each (N = 256: the output layer's logits)
    bool single_ok = transB || (c.N[0] > 192 && rows_wct(c.N[0]) == 4);
#ifdef GEOGCN_F32_NO_ROWS_SINGLE          // A/B build only
    single_ok = false;
#endif
    if (transA || c.panel_w || (c.n_nseg == 1 && c.n_kseg == 1 && !single_ok) || c.M < kRowsMinM) return 0;
    const int64_t kp = cdiv(c.K[0], 16) * 16;
    if (kp != 304 && kp != 256) return 0;
    if (c.n_kseg == 2 && (cdiv(c.K[1], 16) * 16 != kp || c.N[0] > 320)) return 0;      // one accumulator: one column pass
    for (int q = 0; q < c.n_nseg; ++q) {
        if (c.N[q] > 640) return 0;
        const int64_t cols = (int64_t)rows_passes(c.N[q]) * 4 * std::max(rows_wct(c.N[q]), 4) * 16;
        if ((cols - c.N[q]) * 8 > cols) return 0;              // at most an eighth of a pass multiplies zero columns
    }
    return (int)kp;
}
inline size_t rows_slot_bytes(int64_t N, int kp) { return (size_t)4 * rows_passes(N) * kRowsWCT * (kp / 16) * 1024; }
inline size_t rows_ws_bytes(const GemmCall& c, int kp) {
    return c.n_kseg == 2 ? 2 * rows_slot_bytes(c.N[0], kp) : rows_slot_bytes(c.N[0], kp) + (c.n_nseg == 2 ? rows_slot_bytes(c.N[1], kp) : 0);
}

template <int KP>
int launch_rows(const RowsArgs& a, int act, hipStream_t st) {
    constexpr int lds = kRowsBM * (KP + 4) * (int)sizeof(float);
    const int G = (int)std::min<int64_t>((int64_t)kNumCU * 2, a.n_mt);
#define GEOGCN_R(ACT)                                                                                            \
    do {                                                                                                         \
        auto kern = gemm_rows_kernel<KP, ACT>;                                                                   \
        static LdsAttrOnce lds_once;                                                                           \
        if (const int rc_ = lds_once.ensure((const void*)kern, (int)(lds))) return rc_; \
        hipLaunchKernelGGL(kern, dim3((unsigned)G), dim3(TPB), lds, st, a);                                      \
        GEOGCN_LAUNCH_CHECK("gemm_rows_kernel");                                                                 \
    } while (0)
    if (a.gateG && a.postY) {
        auto kern = gemm_rows_kernel<KP, GEOGCN_ACT_NONE, true, true>;
        static LdsAttrOnce lds_once;
        if (const int rc_ = lds_once.ensure((const void*)kern, (int)(lds))) return rc_;
        hipLaunchKernelGGL(kern, dim3((unsigned)G), dim3(TPB), lds, st, a);
        GEOGCN_LAUNCH_CHECK("gemm_rows_kernel");
    } else if (a.gateG) {
        auto kern = gemm_rows_kernel<KP, GEOGCN_ACT_NONE, true>;          // (the gated form has no activation: geogcn_gemm_kcat_gated_f32)
        static LdsAttrOnce lds_once;
        if (const int rc_ = lds_once.ensure((const void*)kern, (int)(lds))) return rc_;
        hipLaunchKernelGGL(kern, dim3((unsigned)G), dim3(TPB), lds, st, a);
        GEOGCN_LAUNCH_CHECK("gemm_rows_kernel");
    } else if (act == GEOGCN_ACT_TANH) GEOGCN_R(GEOGCN_ACT_TANH);
    else if (act == GEOGCN_ACT_SIGMOID) GEOGCN_R(GEOGCN_ACT_SIGMOID);
    else GEOGCN_R(GEOGCN_ACT_NONE);
#undef GEOGCN_R
    return 0;
}

int run_rows(int kp, bool transB, const GemmCall& c, void* ws, hipStream_t st) {
    RowsArgs a{};
    a.M = c.M;
    a.n_mt = (int)cdiv(c.M, kRowsBM);
    a.n_kseg = c.n_kseg;
    a.n_nseg = c.n_nseg;
    a.accumulate = c.accumulate;
    a.gateG = c.gateG; a.ldg = c.ldg; a.gateT = c.gateT; a.ldt = c.ldt;
    a.postY = c.postY; a.ldy = c.ldy; a.postKeep = c.postKeep; a.postF = c.postF; a.postScale = c.postScale;
    float* w = (float*)ws;
    const int nks = kp / 16;
    for (int q = 0; q < 2; ++q) {
        a.A[q] = c.A[q]; a.lda[q] = c.lda[q]; a.K[q] = (int)c.K[q];
        a.C[q] = c.C[q]; a.ldc[q] = c.ldc[q]; a.bias[q] = c.bias[q]; a.N[q] = c.N[q];
        a.act_on[q] = c.act[q] != GEOGCN_ACT_NONE;
        a.passes[q] = c.N[q] > 0 ? rows_passes(c.N[q]) : 0;
        a.wct[q] = c.N[q] > 0 ? std::max(rows_wct(c.N[q]), 4) : kRowsWCT;
    }
    // slot q = N segment | K segment: its weights in fragment order
    const int n_slots = (c.n_nseg == 2 || c.n_kseg == 2) ? 2 : 1;
    for (int q = 0; q < n_slots; ++q) {
        const int64_t N = c.n_kseg == 2 ? c.N[0] : c.N[q];
        const int64_t K = c.n_kseg == 2 ? c.K[q] : c.K[0];
        const int n_tiles = 4 * rows_passes(N) * kRowsWCT;
        const unsigned grid = (unsigned)std::min<int64_t>(cdiv((int64_t)n_tiles * nks * 256, TPB), 1024);
        hipLaunchKernelGGL(prep_b_frag_f32_kernel, dim3(grid), dim3(TPB), 0, st, c.B[q], c.ldb[q], (int)K, (int)N, nks, n_tiles,
                           transB ? 1 : 0, w);
        GEOGCN_LAUNCH_CHECK("prep_b_frag_f32_kernel");
        a.Bf[q] = w;
        w += (size_t)n_tiles * nks * 256;
    }
    const int act = c.act[0] != GEOGCN_ACT_NONE ? c.act[0] : c.act[1];
    if (kp == 304) return launch_rows<304>(a, act, st);
    return launch_rows<256>(a, act, st);
}

int run_call(bool transA, bool transB, const GemmCall& c, void* ws, size_t ws_bytes, hipStream_t st) {
    // fp32-class split-bf16 products where the caller allows them and a kernel takes the shape (gemm_x3.hip); anything else: exact fp32
    if (const int kc = x3_rows_kc(c, transA, transB); kc && ws && aligned16(ws) && ws_bytes >= x3_rows_ws_bytes(c, kc))
        return x3_run_rows(kc, transB, c, ws, st);
    if (const int kp = rows_kp(c, transA, transB); kp && ws && aligned16(ws) && ws_bytes >= rows_ws_bytes(c, kp))
        return run_rows(kp, transB, c, ws, st);          // (too small a workspace -- an older caller: the staged kernel)
    int bm, bn;
    choose_tiles(transA, transB, c.M, c.maxN(), c.n_nseg, bm, bn, c.precision);
    if (transA) return dispatch_tiles<true, false>(bm, bn, c, ws, ws_bytes, st);
    if (transB) return dispatch_tiles<false, true>(bm, bn, c, ws, ws_bytes, st);
    return dispatch_tiles<false, false>(bm, bn, c, ws, ws_bytes, st);
}

template <int BM, int BN, int WM = 2, int WN = 2>
size_t splitk_ws_bytes(int64_t M, int64_t maxN, int n_nseg, int64_t K) {
    const SplitPlan sp = plan_grid<BM, BN, true, false, WM, WN>(M, cdiv(maxN, BN) * n_nseg, K);
    return sp.nsplit <= 1 ? 0 : (size_t)sp.nsplit * (size_t)M * (size_t)(((maxN + 3) & ~(int64_t)3) * n_nseg) * sizeof(float);
}

size_t transA_ws_bytes(int64_t M, int64_t maxN, int n_nseg, int64_t K, int precision) {
    const int bm = pick_tile(M), bn = pick_tile(maxN), wbn = wide_bn(maxN, precision);
    if (wbn == 320) return bm == 160 ? splitk_ws_bytes<160, 320, 2, 4>(M, maxN, n_nseg, K) : splitk_ws_bytes<128, 320, 2, 4>(M, maxN, n_nseg, K);
    if (wbn == 256) return bm == 160 ? splitk_ws_bytes<160, 256, 2, 4>(M, maxN, n_nseg, K) : splitk_ws_bytes<128, 256, 2, 4>(M, maxN, n_nseg, K);
    if (bm == 128 && bn == 128) return splitk_ws_bytes<128, 128>(M, maxN, n_nseg, K);
    if (bm == 128 && bn == 160) return splitk_ws_bytes<128, 160>(M, maxN, n_nseg, K);
    if (bm == 160 && bn == 128) return splitk_ws_bytes<160, 128>(M, maxN, n_nseg, K);
    return splitk_ws_bytes<160, 160>(M, maxN, n_nseg, K);
}

// C = C * (keep * scale) * (1 - Y^2) in place (shapes the whole-rows kernel does not take: the post-operation on its own)
__global__ __launch_bounds__(TPB) void tanh_bwd_post_kernel(int64_t n, int F, int F4, float* __restrict__ C, int64_t ldc,
                                                            const float* __restrict__ Y, int64_t ldy, const uint8_t* __restrict__ keep,
                                                            int64_t keepF, float scale) {
    const int64_t total = n * F4;
    for (int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x; e < total; e += (int64_t)gridDim.x * TPB) {
        const int64_t row = e / F4;
        const int c0 = (int)(e - row * F4) * 4;
        float4 v = *reinterpret_cast<float4*>(C + row * ldc + c0);
        const float4 y = *reinterpret_cast<const float4*>(Y + row * ldy + c0);
        float* vv = reinterpret_cast<float*>(&v);
        const float* yy = reinterpret_cast<const float*>(&y);
#pragma unroll
        for (int i = 0; i < 4; ++i) vv[i] = (c0 + i < F) ? masked_tanh_bwd(vv[i], (float)keep[row * keepF + c0 + i], scale, yy[i]) : 0.f;
        *reinterpret_cast<float4*>(C + row * ldc + c0) = v;
    }
}

bool ld_ok(const void* p, int64_t ld) { return ld % 4 == 0 && aligned16(p); }

}  // namespace
}  // namespace geogcn

using namespace geogcn;

extern "C" {

size_t geogcn_gemm_workspace_bytes(int32_t transA, int32_t transB, int64_t M, int64_t N, int64_t K,
                                   int32_t precision) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    if (!transA && precision == GEOGCN_GEMM_F32) {      // the whole-rows kernel's fragment-ordered weights (0: the staged kernel)
        GemmCall c{};
        c.M = M; c.n_nseg = 1; c.n_kseg = 1; c.N[0] = N; c.K[0] = K;
        const int kp = rows_kp(c, false, transB != 0);
        return kp ? rows_ws_bytes(c, kp) : 0;
    }
    if (!transA && precision == GEOGCN_GEMM_BF16X3) {
        // the split-bf16 whole-rows kernel's weights; the exact whole-rows kernel's (the shapes gemm_x3.hip does not take run exact
        // fp32); the staged split-bf16 kernel's planes (panel outputs)
        GemmCall c{};
        c.M = M; c.n_nseg = 1; c.n_kseg = 1; c.N[0] = N; c.K[0] = K; c.precision = precision;
        const int kc = x3_rows_kc(c, false, transB != 0);
        const int kp = rows_kp(c, false, transB != 0);
        return std::max(std::max(kc ? x3_rows_ws_bytes(c, kc) : (size_t)0, kp ? rows_ws_bytes(c, kp) : (size_t)0),
                        gemm_bf16_workspace_bytes(precision, N, K));
    }
    if (!transA) return gemm_bf16_workspace_bytes(precision, N, K);
    if (precision == GEOGCN_GEMM_BF16) {
        const size_t h = gemm_bf16_tn_workspace_bytes(M, N, K);      // 0: shape left to the fp32 kernel
        if (h) return h;
    }
    return transA_ws_bytes(M, N, 1, K, precision);
}

size_t geogcn_gemm_dual_workspace_bytes(int32_t transA, int64_t M, int64_t N0, int64_t N1, int64_t K, int32_t precision) {
    if (M <= 0 || N0 <= 0 || N1 <= 0 || K <= 0) return 0;
    if (transA && precision == GEOGCN_GEMM_BF16)          // (round 6) the bf16 configuration's two weight gradients in one launch, or its two launches
        return std::max(gemm_bf16_tn_dual_workspace_bytes(M, N0, N1, K),
                        std::max(geogcn_gemm_workspace_bytes(1, 0, M, N0, K, precision), geogcn_gemm_workspace_bytes(1, 0, M, N1, K, precision)));
    if (!transA) {          // the whole-rows kernels' fragment-ordered weights (0: the call runs on the staged kernel)
        GemmCall c{};
        c.M = M; c.n_nseg = 2; c.n_kseg = 1; c.N[0] = N0; c.N[1] = N1; c.K[0] = K; c.precision = precision;
        if (const int kc = x3_rows_kc(c, false, false)) return x3_rows_ws_bytes(c, kc);
        const int kp = rows_kp(c, false);
        return kp ? rows_ws_bytes(c, kp) : 0;
    }
    return transA_ws_bytes(M, std::max(N0, N1), 2, K, precision);
}

size_t geogcn_gemm_kcat_workspace_bytes(int32_t transB, int64_t M, int64_t N, int64_t K0, int64_t K1, int32_t precision) {
    if (M <= 0 || N <= 0 || K0 <= 0 || K1 <= 0) return 0;
    if (precision == GEOGCN_GEMM_BF16) return gemm_bf16_kcat_workspace_bytes(N, K0, K1);      // (also covers the two separate launches of other shapes)
    GemmCall c{};
    c.M = M; c.n_nseg = 1; c.n_kseg = 2; c.N[0] = N; c.K[0] = K0; c.K[1] = K1; c.precision = precision;
    if (const int kc = x3_rows_kc(c, false, transB != 0)) return x3_rows_ws_bytes(c, kc);
    const int kp = rows_kp(c, false);
    return kp ? rows_ws_bytes(c, kp) : 0;
}

static int gemm_entry(const char* fn, int32_t transA, int32_t transB, int64_t M, int64_t N, int64_t K, const float* A,
                      int64_t lda, const float* B, int64_t ldb, void* Cv, int64_t ldc, int c_bf16, const float* bias,
                      int32_t act, int32_t accumulate, int32_t precision, void* ws, size_t ws_bytes, void* stream,
                      int panel_w = 0, int64_t panel_R = 0) {
    GEOGCN_REQUIRE(M >= 0 && N >= 0 && K >= 0, GEOGCN_E_SIZE, "%s: negative size", fn);
    GEOGCN_REQUIRE(precision >= GEOGCN_GEMM_F32 && precision <= GEOGCN_GEMM_BF16, GEOGCN_E_ARG,
                   "%s: unknown precision %d", fn, precision);
    if (M == 0 || N == 0) return 0;
    GEOGCN_REQUIRE(Cv && (K == 0 || (A && B)), GEOGCN_E_NULL, "%s: null pointer", fn);
    GEOGCN_REQUIRE(act >= GEOGCN_ACT_NONE && act <= GEOGCN_ACT_SIGMOID, GEOGCN_E_ARG, "%s: unknown act %d", fn, act);
    GEOGCN_REQUIRE(!(transA && transB), GEOGCN_E_ARG, "%s: transA && transB not supported", fn);
    const int64_t a_cols = transA ? M : K, b_cols = transB ? K : N;
    GEOGCN_REQUIRE(lda >= a_cols && ldb >= b_cols && ldc >= N, GEOGCN_E_SIZE,
                   "%s: leading dimension too small (lda=%lld ldb=%lld ldc=%lld)", fn, (long long)lda,
                   (long long)ldb, (long long)ldc);
    GEOGCN_REQUIRE(lda % 4 == 0 && ldb % 4 == 0 && ldc % 4 == 0 && aligned16(A) && aligned16(B) && aligned16(Cv),
                   GEOGCN_E_ALIGN,
                   "%s: A, B, C need 16-byte aligned bases and ld %% 4 == 0 (lda=%lld ldb=%lld ldc=%lld)", fn,
                   (long long)lda, (long long)ldb, (long long)ldc);
    hipStream_t st = (hipStream_t)stream;
    if (c_bf16) {
        GEOGCN_REQUIRE(!transA && precision == GEOGCN_GEMM_BF16 && !accumulate && K > 0, GEOGCN_E_ARG,
                       "%s: a bf16 C needs transA = 0, precision = GEOGCN_GEMM_BF16, accumulate = 0", fn);
        GEOGCN_REQUIRE(ldc % 8 == 0 && ldc >= ((N + 7) & ~(int64_t)7), GEOGCN_E_ALIGN,
                       "%s: a bf16 C needs ldc %% 8 == 0 and >= roundup8(N) (ldc=%lld)", fn, (long long)ldc);
    }
    if (panel_w) {
        GEOGCN_REQUIRE(!transA && !accumulate && K > 0 && panel_w % (c_bf16 ? 8 : 4) == 0 && panel_R >= M, GEOGCN_E_ARG,
                       "%s: panel output needs transA = 0, accumulate = 0, K > 0, panel_w %% %d == 0, panel_R >= M", fn,
                       c_bf16 ? 8 : 4);
    }
    if (K == 0) {
        // an empty reduction (a rank that owns no rows: dW = H^T.dZ over zero nodes): the product is the zero matrix
        if (accumulate) return 0;
        GEOGCN_REQUIRE(!bias && act == GEOGCN_ACT_NONE, GEOGCN_E_ARG, "%s: K = 0 with a bias / activation", fn);
        return zero_rows_async((float*)Cv, M, (N + 3) & ~(int64_t)3, ldc, st);
    }
    // GEOGCN_GEMM_BF16X3 is a permission: the split-bf16 kernels of gemm_x3.hip where they take the shape (run_call below), the staged
    // split-bf16 kernel for panel outputs, exact fp32 for everything else -- so that a fused launch and its separate launches stay
    // bit-identical at every size
    if (!transA && precision == GEOGCN_GEMM_BF16X3 && panel_w && !c_bf16) {
        // (round 6) the whole-rows split-bf16 kernel writes the panels itself where it takes the shape (the TwitterUS-size ranks' H . W);
        // smaller calls: the staged split-bf16 kernel below, as before
        GemmCall c{};
        c.M = M; c.n_nseg = 1; c.n_kseg = 1;
        c.A[0] = A; c.lda[0] = lda; c.B[0] = B; c.ldb[0] = ldb; c.C[0] = (float*)Cv; c.ldc[0] = ldc; c.bias[0] = bias;
        c.N[0] = N; c.K[0] = K; c.act[0] = act; c.act[1] = GEOGCN_ACT_NONE; c.accumulate = 0;
        c.panel_w = panel_w; c.panel_R = panel_R; c.precision = precision;
        if (const int kc = x3_rows_kc(c, false, transB != 0); kc && ws && aligned16(ws) && ws_bytes >= x3_rows_ws_bytes(c, kc))
            return x3_run_rows(kc, transB != 0, c, ws, st);
    }
    if (!transA && (precision == GEOGCN_GEMM_BF16 || (precision == GEOGCN_GEMM_BF16X3 && panel_w)))
        return gemm_bf16_dispatch(precision, transB, M, N, K, A, lda, B, ldb, Cv, ldc, c_bf16, bias, act, accumulate, ws,
                                  ws_bytes, st, panel_w, panel_R);
    float* C = (float*)Cv;
    if (transA && precision == GEOGCN_GEMM_BF16) {
        const int rc = gemm_bf16_tn_dispatch(M, N, K, A, lda, B, ldb, C, ldc, bias, act, accumulate, ws, ws_bytes, st);
        if (rc != 1) return rc;          // 1 = shape not handled by the bf16 kernel: exact fp32 below
    }
    GemmCall c{};
    c.M = M; c.n_nseg = 1; c.n_kseg = 1;
    c.A[0] = A; c.lda[0] = lda; c.B[0] = B; c.ldb[0] = ldb; c.C[0] = C; c.ldc[0] = ldc; c.bias[0] = bias;
    c.N[0] = N; c.K[0] = K; c.act[0] = act; c.act[1] = GEOGCN_ACT_NONE; c.accumulate = accumulate;
    c.panel_w = panel_w; c.panel_R = panel_R;
    c.precision = precision == GEOGCN_GEMM_BF16X3 ? precision : GEOGCN_GEMM_F32;      // (transA: split-bf16 slabs where a kernel takes the tile)
    return run_call(transA != 0, transB != 0, c, ws, ws_bytes, st);
}

int geogcn_gemm_f32(int32_t transA, int32_t transB, int64_t M, int64_t N, int64_t K, const float* A,
                    int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, const float* bias,
                    int32_t act, int32_t accumulate, int32_t precision, void* ws, size_t ws_bytes, void* stream) {
    return gemm_entry("gemm_f32", transA, transB, M, N, K, A, lda, B, ldb, C, ldc, 0, bias, act, accumulate, precision, ws,
                      ws_bytes, stream);
}

int geogcn_gemm_f32_bf16c(int32_t transB, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B,
                          int64_t ldb, uint16_t* C, int64_t ldc, const float* bias, int32_t act, void* ws,
                          size_t ws_bytes, void* stream) {
    return gemm_entry("gemm_f32_bf16c", 0, transB, M, N, K, A, lda, B, ldb, C, ldc, 1, bias, act, 0, GEOGCN_GEMM_BF16, ws,
                      ws_bytes, stream);
}

// C as feature panels [W][R][wp] (fp32, or bf16 with precision = GEOGCN_GEMM_BF16 and c_bf16 = 1)
int geogcn_gemm_panels_f32(int32_t transB, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B,
                           int64_t ldb, void* panels, int64_t R, int32_t W, int32_t wp, int32_t c_bf16, const float* bias,
                           int32_t act, int32_t precision, void* ws, size_t ws_bytes, void* stream) {
    GEOGCN_REQUIRE(W > 0 && wp > 0 && (int64_t)W * wp >= N && R >= M, GEOGCN_E_SIZE,
                   "gemm_panels_f32: need W * wp >= N and R >= M (W=%d wp=%d N=%lld R=%lld M=%lld)", W, wp, (long long)N,
                   (long long)R, (long long)M);
    GEOGCN_REQUIRE(!c_bf16 || precision == GEOGCN_GEMM_BF16, GEOGCN_E_ARG, "gemm_panels_f32: bf16 panels need precision = BF16");
    // (ldc only has to pass the row-major checks: the panel addressing replaces it)
    const int64_t ldc = c_bf16 ? ((N + 7) & ~(int64_t)7) : ((N + 3) & ~(int64_t)3);
    return gemm_entry("gemm_panels_f32", 0, transB, M, N, K, A, lda, B, ldb, panels, ldc, c_bf16, bias, act, 0, precision, ws,
                      ws_bytes, stream, wp, R);
}

// (C0, C1) = (act0(op(A).B0 + bias0), act1(op(A).B1 + bias1)) in one launch, exact fp32
int geogcn_gemm_dual_f32(int32_t transA, int64_t M, int64_t N0, int64_t N1, int64_t K, const float* A, int64_t lda,
                         const float* B0, int64_t ldb0, const float* B1, int64_t ldb1, float* C0, int64_t ldc0,
                         float* C1, int64_t ldc1, const float* bias0, int32_t act0, const float* bias1, int32_t act1,
                         int32_t precision, void* ws, size_t ws_bytes, void* stream) {
    const char* fn = "gemm_dual_f32";
    GEOGCN_REQUIRE(precision == GEOGCN_GEMM_F32 || precision == GEOGCN_GEMM_BF16X3 || (precision == GEOGCN_GEMM_BF16 && transA), GEOGCN_E_ARG,
                   "%s: precision must be F32 or BF16X3 (or BF16 with transA = 1; the bf16 forward pair is geogcn_gemm_dual_bf16) (%d)", fn, precision);
    GEOGCN_REQUIRE(M >= 0 && N0 >= 0 && N1 >= 0 && K >= 0, GEOGCN_E_SIZE, "%s: negative size", fn);
    if (M == 0 || (N0 == 0 && N1 == 0)) return 0;
    GEOGCN_REQUIRE(N0 > 0 && N1 > 0, GEOGCN_E_SIZE, "%s: both products need columns (N0=%lld N1=%lld)", fn, (long long)N0,
                   (long long)N1);
    GEOGCN_REQUIRE(C0 && C1 && (K == 0 || (A && B0 && B1)), GEOGCN_E_NULL, "%s: null pointer", fn);
    GEOGCN_REQUIRE(act0 >= GEOGCN_ACT_NONE && act0 <= GEOGCN_ACT_SIGMOID && act1 >= GEOGCN_ACT_NONE && act1 <= GEOGCN_ACT_SIGMOID,
                   GEOGCN_E_ARG, "%s: unknown act", fn);
    GEOGCN_REQUIRE(act0 == GEOGCN_ACT_NONE || act1 == GEOGCN_ACT_NONE || act0 == act1, GEOGCN_E_ARG,
                   "%s: two different non-linear epilogues (%d, %d) in one launch are not supported", fn, act0, act1);
    const int64_t a_cols = transA ? M : K;
    GEOGCN_REQUIRE(lda >= a_cols && ldb0 >= N0 && ldb1 >= N1 && ldc0 >= N0 && ldc1 >= N1, GEOGCN_E_SIZE,
                   "%s: leading dimension too small", fn);
    GEOGCN_REQUIRE(ld_ok(A, lda) && ld_ok(B0, ldb0) && ld_ok(B1, ldb1) && ld_ok(C0, ldc0) && ld_ok(C1, ldc1), GEOGCN_E_ALIGN,
                   "%s: operands need 16-byte aligned bases and ld %% 4 == 0", fn);
    hipStream_t st = (hipStream_t)stream;
    if (K == 0) {
        GEOGCN_REQUIRE(!bias0 && !bias1 && act0 == GEOGCN_ACT_NONE && act1 == GEOGCN_ACT_NONE, GEOGCN_E_ARG,
                       "%s: K = 0 with a bias / activation", fn);
        const int rc = zero_rows_async(C0, M, (N0 + 3) & ~(int64_t)3, ldc0, st);
        return rc ? rc : zero_rows_async(C1, M, (N1 + 3) & ~(int64_t)3, ldc1, st);
    }
    if (precision == GEOGCN_GEMM_BF16) {          // transA: (dW0, dW1) = A^T . [B0 | B1], bf16 products
        GEOGCN_REQUIRE(!bias0 && !bias1 && act0 == GEOGCN_ACT_NONE && act1 == GEOGCN_ACT_NONE, GEOGCN_E_ARG, "%s: the bf16 A^T . [B0 | B1] has no epilogue", fn);
        const int rc = gemm_bf16_tn_dual_dispatch(M, N0, N1, K, A, lda, B0, ldb0, B1, ldb1, C0, ldc0, C1, ldc1, ws, ws_bytes, st);
        if (rc != 1) return rc;
        if (const int r = geogcn_gemm_f32(1, 0, M, N0, K, A, lda, B0, ldb0, C0, ldc0, nullptr, GEOGCN_ACT_NONE, 0, precision, ws, ws_bytes, stream)) return r;
        return geogcn_gemm_f32(1, 0, M, N1, K, A, lda, B1, ldb1, C1, ldc1, nullptr, GEOGCN_ACT_NONE, 0, precision, ws, ws_bytes, stream);
    }
    GemmCall c{};
    c.M = M; c.n_nseg = 2; c.n_kseg = 1;
    c.A[0] = A; c.lda[0] = lda;
    c.B[0] = B0; c.ldb[0] = ldb0; c.B[1] = B1; c.ldb[1] = ldb1;
    c.C[0] = C0; c.ldc[0] = ldc0; c.C[1] = C1; c.ldc[1] = ldc1;
    c.bias[0] = bias0; c.bias[1] = bias1;
    c.N[0] = N0; c.N[1] = N1; c.K[0] = K; c.act[0] = act0; c.act[1] = act1; c.precision = precision;
    return run_call(transA != 0, false, c, ws, ws_bytes, st);
}

// (round 6) the bf16 configuration's k-concatenated product: one launch of the bf16 whole-rows kernel where it takes the shape, else the
// two launches the reverse sweep made until now (the first writes -- with the carry --, the second accumulates)
static int kcat_bf16(int32_t transB, int64_t M, int64_t N, int64_t K0, int64_t K1, const float* A0, int64_t lda0, const float* B0, int64_t ldb0,
                     const float* A1, int64_t lda1, const float* B1, int64_t ldb1, float* C, int64_t ldc, int accumulate, const GateOps* gate,
                     void* ws, size_t ws_bytes, hipStream_t st) {
    const int rc = gemm_bf16_kcat_dispatch(transB, M, N, K0, K1, A0, lda0, B0, ldb0, A1, lda1, B1, ldb1, C, ldc, accumulate, ws, ws_bytes, st, gate);
    if (rc != 1) return rc;
    if (const int r = gemm_bf16_dispatch(GEOGCN_GEMM_BF16, transB, M, N, K0, A0, lda0, B0, ldb0, C, ldc, 0, nullptr, GEOGCN_ACT_NONE, accumulate,
                                         ws, ws_bytes, st, 0, 0, gate))
        return r;
    return gemm_bf16_dispatch(GEOGCN_GEMM_BF16, transB, M, N, K1, A1, lda1, B1, ldb1, C, ldc, 0, nullptr, GEOGCN_ACT_NONE, 1, ws, ws_bytes, st, 0, 0,
                              nullptr);
}

// C = A0.op(B0) + A1.op(B1) [+ C]: one accumulator over both reductions, exact fp32
int geogcn_gemm_kcat_f32(int32_t transB, int64_t M, int64_t N, int64_t K0, int64_t K1, const float* A0, int64_t lda0,
                         const float* B0, int64_t ldb0, const float* A1, int64_t lda1, const float* B1, int64_t ldb1,
                         float* C, int64_t ldc, int32_t accumulate, int32_t precision, void* ws, size_t ws_bytes, void* stream) {
    const char* fn = "gemm_kcat_f32";
    GEOGCN_REQUIRE(precision >= GEOGCN_GEMM_F32 && precision <= GEOGCN_GEMM_BF16, GEOGCN_E_ARG, "%s: unknown precision %d", fn, precision);
    GEOGCN_REQUIRE(M >= 0 && N >= 0 && K0 > 0 && K1 > 0, GEOGCN_E_SIZE, "%s: bad sizes", fn);
    if (M == 0 || N == 0) return 0;
    GEOGCN_REQUIRE(A0 && A1 && B0 && B1 && C, GEOGCN_E_NULL, "%s: null pointer", fn);
    const int64_t b0_cols = transB ? K0 : N, b1_cols = transB ? K1 : N;
    GEOGCN_REQUIRE(lda0 >= K0 && lda1 >= K1 && ldb0 >= b0_cols && ldb1 >= b1_cols && ldc >= N, GEOGCN_E_SIZE,
                   "%s: leading dimension too small", fn);
    GEOGCN_REQUIRE(ld_ok(A0, lda0) && ld_ok(A1, lda1) && ld_ok(B0, ldb0) && ld_ok(B1, ldb1) && ld_ok(C, ldc), GEOGCN_E_ALIGN,
                   "%s: operands need 16-byte aligned bases and ld %% 4 == 0", fn);
    if (precision == GEOGCN_GEMM_BF16)
        return kcat_bf16(transB, M, N, K0, K1, A0, lda0, B0, ldb0, A1, lda1, B1, ldb1, C, ldc, accumulate, nullptr, ws, ws_bytes, (hipStream_t)stream);
    GemmCall c{};
    c.M = M; c.n_nseg = 1; c.n_kseg = 2;
    c.A[0] = A0; c.lda[0] = lda0; c.A[1] = A1; c.lda[1] = lda1;
    c.B[0] = B0; c.ldb[0] = ldb0; c.B[1] = B1; c.ldb[1] = ldb1;
    c.C[0] = C; c.ldc[0] = ldc;
    c.N[0] = N; c.K[0] = K0; c.K[1] = K1; c.accumulate = accumulate; c.precision = precision;
    return run_call(false, transB != 0, c, ws, ws_bytes, (hipStream_t)stream);
}

size_t geogcn_gemm_dual_bf16_workspace_bytes(int64_t M, int64_t N0, int64_t N1, int64_t K) {
    if (M <= 0 || N0 <= 0 || N1 <= 0 || K <= 0) return 0;
    return gemm_bf16_dual_workspace_bytes(N0, N1, K);
}

int geogcn_gemm_dual_bf16(int64_t M, int64_t N0, int64_t N1, int64_t K, const float* A, int64_t lda, const float* B0, int64_t ldb0,
                          const float* B1, int64_t ldb1, void* C0, int64_t ldc0, int32_t c0_bf16, float* C1, int64_t ldc1,
                          const float* bias1, int32_t act1, void* ws, size_t ws_bytes, void* stream) {
    const char* fn = "gemm_dual_bf16";
    GEOGCN_REQUIRE(M >= 0 && N0 > 0 && N1 > 0 && K > 0, GEOGCN_E_SIZE, "%s: bad sizes", fn);
    GEOGCN_REQUIRE(act1 >= GEOGCN_ACT_NONE && act1 <= GEOGCN_ACT_SIGMOID, GEOGCN_E_ARG, "%s: unknown act %d", fn, act1);
    if (M == 0) return 0;
    GEOGCN_REQUIRE(A && B0 && B1 && C0 && C1, GEOGCN_E_NULL, "%s: null pointer", fn);
    GEOGCN_REQUIRE(lda >= K && ldb0 >= N0 && ldb1 >= N1 && ldc0 >= N0 && ldc1 >= N1, GEOGCN_E_SIZE, "%s: leading dimension too small", fn);
    GEOGCN_REQUIRE(ld_ok(A, lda) && ld_ok(B0, ldb0) && ld_ok(B1, ldb1) && ld_ok(C1, ldc1) && aligned16(C0), GEOGCN_E_ALIGN,
                   "%s: operands need 16-byte aligned bases and ld %% 4 == 0", fn);
    if (c0_bf16)
        GEOGCN_REQUIRE(ldc0 % 8 == 0 && ldc0 >= ((N0 + 7) & ~(int64_t)7), GEOGCN_E_ALIGN, "%s: a bf16 C0 needs ldc0 %% 8 == 0 and >= roundup8(N0)", fn);
    else
        GEOGCN_REQUIRE(ldc0 % 4 == 0, GEOGCN_E_ALIGN, "%s: ldc0 %% 4 != 0", fn);
    return gemm_bf16_dual_dispatch(M, N0, N1, K, A, lda, B0, ldb0, B1, ldb1, C0, ldc0, c0_bf16, C1, ldc1, bias1, act1, ws, ws_bytes,
                                   (hipStream_t)stream);
}

int geogcn_gemm_gated_f32(int32_t transB, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B,
                          int64_t ldb, float* C, int64_t ldc, const float* G, int64_t ldg, const float* T, int64_t ldt,
                          int32_t precision, void* ws, size_t ws_bytes, void* stream) {
    const char* fn = "gemm_gated_f32";
    GEOGCN_REQUIRE(M >= 0 && N >= 0 && K > 0, GEOGCN_E_SIZE, "%s: bad sizes", fn);
    GEOGCN_REQUIRE(precision >= GEOGCN_GEMM_F32 && precision <= GEOGCN_GEMM_BF16, GEOGCN_E_ARG, "%s: unknown precision %d", fn, precision);
    if (M == 0 || N == 0) return 0;
    GEOGCN_REQUIRE(A && B && C && G && T, GEOGCN_E_NULL, "%s: null pointer", fn);
    const int64_t b_cols = transB ? K : N, n4 = (N + 3) & ~(int64_t)3;
    GEOGCN_REQUIRE(lda >= K && ldb >= b_cols && ldc >= N && ldg >= n4 && ldt >= n4, GEOGCN_E_SIZE, "%s: leading dimension too small", fn);
    GEOGCN_REQUIRE(ld_ok(A, lda) && ld_ok(B, ldb) && ld_ok(C, ldc) && ld_ok(G, ldg) && ld_ok(T, ldt), GEOGCN_E_ALIGN,
                   "%s: operands need 16-byte aligned bases and ld %% 4 == 0", fn);
    GEOGCN_REQUIRE(C != G && C != T, GEOGCN_E_ARG, "%s: C must not alias G or T", fn);
    hipStream_t st = (hipStream_t)stream;
    if (precision == GEOGCN_GEMM_BF16X3) {
        GemmCall c{};
        c.M = M; c.n_nseg = 1; c.n_kseg = 1;
        c.A[0] = A; c.lda[0] = lda; c.B[0] = B; c.ldb[0] = ldb; c.C[0] = C; c.ldc[0] = ldc;
        c.N[0] = N; c.K[0] = K; c.act[0] = c.act[1] = GEOGCN_ACT_NONE; c.precision = precision;
        if (const int kc = x3_rows_kc(c, false, transB != 0); kc && ws && aligned16(ws) && ws_bytes >= x3_rows_ws_bytes(c, kc)) {
            c.gateG = G; c.ldg = ldg; c.gateT = T; c.ldt = ldt;
            return x3_run_rows(kc, transB != 0, c, ws, st);
        }
    }
    if (precision == GEOGCN_GEMM_BF16) {
        const GateOps gate{G, ldg, T, ldt};
        return gemm_bf16_dispatch(precision, transB, M, N, K, A, lda, B, ldb, C, ldc, 0, nullptr, GEOGCN_ACT_NONE, 0, ws, ws_bytes, st,
                                  0, 0, &gate);
    }
    GemmCall c{};
    c.M = M; c.n_nseg = 1; c.n_kseg = 1;
    c.A[0] = A; c.lda[0] = lda; c.B[0] = B; c.ldb[0] = ldb; c.C[0] = C; c.ldc[0] = ldc;
    c.N[0] = N; c.K[0] = K; c.act[0] = c.act[1] = GEOGCN_ACT_NONE;
    if (const int kp = rows_kp(c, false, transB != 0); kp && ws && aligned16(ws) && ws_bytes >= rows_ws_bytes(c, kp)) {
        c.gateG = G; c.ldg = ldg; c.gateT = T; c.ldt = ldt;
        return run_rows(kp, transB != 0, c, ws, st);
    }
    if (const int rc = geogcn_gate_carry_f32(M, (int32_t)N, G, ldg, T, ldt, C, ldc, stream)) return rc;
    c.accumulate = 1;
    return run_call(false, transB != 0, c, ws, ws_bytes, st);
}

static int kcat_gated_impl(const char* fn, int32_t transB, int64_t M, int64_t N, int64_t K0, int64_t K1, const float* A0, int64_t lda0,
                           const float* B0, int64_t ldb0, const float* A1, int64_t lda1, const float* B1, int64_t ldb1,
                           float* C, int64_t ldc, const float* G, int64_t ldg, const float* T, int64_t ldt, const float* Y, int64_t ldy,
                           const uint8_t* keep, int64_t keepF, float scale, int32_t precision, void* ws, size_t ws_bytes, void* stream);

int geogcn_gemm_kcat_gated_f32(int32_t transB, int64_t M, int64_t N, int64_t K0, int64_t K1, const float* A0, int64_t lda0,
                               const float* B0, int64_t ldb0, const float* A1, int64_t lda1, const float* B1, int64_t ldb1,
                               float* C, int64_t ldc, const float* G, int64_t ldg, const float* T, int64_t ldt, int32_t precision,
                               void* ws, size_t ws_bytes, void* stream) {
    return kcat_gated_impl("gemm_kcat_gated_f32", transB, M, N, K0, K1, A0, lda0, B0, ldb0, A1, lda1, B1, ldb1, C, ldc, G, ldg, T, ldt,
                           nullptr, 0, nullptr, 0, 0.f, precision, ws, ws_bytes, stream);
}

int geogcn_gemm_kcat_gated_tanhbwd_f32(int32_t transB, int64_t M, int64_t N, int64_t K0, int64_t K1, const float* A0, int64_t lda0,
                                       const float* B0, int64_t ldb0, const float* A1, int64_t lda1, const float* B1, int64_t ldb1,
                                       float* C, int64_t ldc, const float* G, int64_t ldg, const float* T, int64_t ldt,
                                       const float* Y, int64_t ldy, const uint8_t* keep, int64_t keepF, float scale, int32_t precision,
                                       void* ws, size_t ws_bytes, void* stream) {
    const char* fn = "gemm_kcat_gated_tanhbwd_f32";
    GEOGCN_REQUIRE(M >= 0 && N >= 0, GEOGCN_E_SIZE, "%s: bad sizes", fn);
    if (M > 0 && N > 0) {
        GEOGCN_REQUIRE(Y && keep, GEOGCN_E_NULL, "%s: null pointer", fn);
        GEOGCN_REQUIRE(ld_ok(Y, ldy) && ldy >= ((N + 3) & ~(int64_t)3) && keepF >= N && keepF % 4 == 0 && (uintptr_t)keep % 4 == 0,
                       GEOGCN_E_ALIGN, "%s: Y needs a 16-byte aligned base and ld %% 4 == 0, the keep mask a pitch %% 4 == 0 >= N", fn);
        GEOGCN_REQUIRE((const float*)C != Y, GEOGCN_E_ARG, "%s: C must not alias Y", fn);
    }
    return kcat_gated_impl(fn, transB, M, N, K0, K1, A0, lda0, B0, ldb0, A1, lda1, B1, ldb1, C, ldc, G, ldg, T, ldt, Y, ldy, keep, keepF,
                           scale, precision, ws, ws_bytes, stream);
}

static int kcat_gated_impl(const char* fn, int32_t transB, int64_t M, int64_t N, int64_t K0, int64_t K1, const float* A0, int64_t lda0,
                           const float* B0, int64_t ldb0, const float* A1, int64_t lda1, const float* B1, int64_t ldb1,
                           float* C, int64_t ldc, const float* G, int64_t ldg, const float* T, int64_t ldt, const float* Y, int64_t ldy,
                           const uint8_t* keep, int64_t keepF, float scale, int32_t precision, void* ws, size_t ws_bytes, void* stream) {
    GEOGCN_REQUIRE(M >= 0 && N >= 0 && K0 > 0 && K1 > 0, GEOGCN_E_SIZE, "%s: bad sizes", fn);
    GEOGCN_REQUIRE(precision >= GEOGCN_GEMM_F32 && precision <= GEOGCN_GEMM_BF16, GEOGCN_E_ARG, "%s: unknown precision %d", fn, precision);
    if (M == 0 || N == 0) return 0;
    GEOGCN_REQUIRE(A0 && A1 && B0 && B1 && C && G && T, GEOGCN_E_NULL, "%s: null pointer", fn);
    const int64_t b0_cols = transB ? K0 : N, b1_cols = transB ? K1 : N, n4 = (N + 3) & ~(int64_t)3;
    GEOGCN_REQUIRE(lda0 >= K0 && lda1 >= K1 && ldb0 >= b0_cols && ldb1 >= b1_cols && ldc >= N && ldg >= n4 && ldt >= n4, GEOGCN_E_SIZE,
                   "%s: leading dimension too small", fn);
    GEOGCN_REQUIRE(ld_ok(A0, lda0) && ld_ok(A1, lda1) && ld_ok(B0, ldb0) && ld_ok(B1, ldb1) && ld_ok(C, ldc) && ld_ok(G, ldg) &&
                       ld_ok(T, ldt),
                   GEOGCN_E_ALIGN, "%s: operands need 16-byte aligned bases and ld %% 4 == 0", fn);
    GEOGCN_REQUIRE(C != G && C != T, GEOGCN_E_ARG, "%s: C must not alias G or T", fn);
    if (precision == GEOGCN_GEMM_BF16) {
        const GateOps gate{G, ldg, T, ldt};
        if (const int rc = kcat_bf16(transB, M, N, K0, K1, A0, lda0, B0, ldb0, A1, lda1, B1, ldb1, C, ldc, 0, &gate, ws, ws_bytes, (hipStream_t)stream))
            return rc;
        if (Y) {
            const int F4 = (int)((N + 3) / 4);
            const int64_t blocks = std::min<int64_t>(cdiv(M * F4, TPB), (int64_t)kNumCU * 16);
            hipLaunchKernelGGL(tanh_bwd_post_kernel, dim3((unsigned)blocks), dim3(TPB), 0, (hipStream_t)stream, M, (int)N, F4, C, ldc, Y, ldy, keep, keepF, scale);
            GEOGCN_LAUNCH_CHECK("tanh_bwd_post_kernel");
        }
        return 0;
    }
    GemmCall c{};
    c.M = M; c.n_nseg = 1; c.n_kseg = 2;
    c.A[0] = A0; c.lda[0] = lda0; c.A[1] = A1; c.lda[1] = lda1;
    c.B[0] = B0; c.ldb[0] = ldb0; c.B[1] = B1; c.ldb[1] = ldb1;
    c.C[0] = C; c.ldc[0] = ldc;
    c.N[0] = N; c.K[0] = K0; c.K[1] = K1; c.precision = precision;
    hipStream_t st = (hipStream_t)stream;
    if (const int kc = x3_rows_kc(c, false, transB != 0); kc && ws && aligned16(ws) && ws_bytes >= x3_rows_ws_bytes(c, kc)) {
        c.accumulate = 0;
        c.gateG = G; c.ldg = ldg; c.gateT = T; c.ldt = ldt;
        c.postY = Y; c.ldy = ldy; c.postKeep = keep; c.postF = keepF; c.postScale = scale;
        return x3_run_rows(kc, transB != 0, c, ws, st);
    }
    if (const int kp = rows_kp(c, false, transB != 0); kp && ws && aligned16(ws) && ws_bytes >= rows_ws_bytes(c, kp)) {
        c.accumulate = 0;
        c.gateG = G; c.ldg = ldg; c.gateT = T; c.ldt = ldt;
        c.postY = Y; c.ldy = ldy; c.postKeep = keep; c.postF = keepF; c.postScale = scale;
        return run_rows(kp, transB != 0, c, ws, st);
    }
    // any other shape: the carry written first, the two products accumulated onto it (the same values in the same order)
    if (const int rc = geogcn_gate_carry_f32(M, (int32_t)N, G, ldg, T, ldt, C, ldc, stream)) return rc;
    c.accumulate = 1;
    if (const int rc = run_call(false, transB != 0, c, ws, ws_bytes, st)) return rc;
    if (Y) {
        const int F4 = (int)((N + 3) / 4);
        const int64_t blocks = std::min<int64_t>(cdiv(M * F4, TPB), (int64_t)kNumCU * 16);
        hipLaunchKernelGGL(tanh_bwd_post_kernel, dim3((unsigned)blocks), dim3(TPB), 0, st, M, (int)N, F4, C, ldc, Y, ldy, keep, keepF, scale);
        GEOGCN_LAUNCH_CHECK("tanh_bwd_post_kernel");
    }
    return 0;
}

}  // extern "C"
