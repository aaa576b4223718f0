// The fp32-class split-bf16 ("bf16x3") contraction on the two GEMM structures that carry the GCN step at the TwitterUS size (gfx950):
//   reference: T.dot(input, W) gcnmodel.py:126,149; the DenseLayer gate gcnmodel.py:285; and the Gemm ops Theano autodiff derives
//   (dW = H^T.dZ, dH = dZ.W^T) -- the same products gemm.hip forms with the exact fp32 MFMA.
// Every fp32 value is split EXACTLY into three bf16 terms (x = x1 + x2 + x3, 8 + 8 + 8 significand bits, round-to-nearest residuals that
// are exact in fp32) and a product a.b is formed from the six largest cross terms a1b1 + a1b2 + a2b1 + a2b2 + a1b3 + a3b1 on
// v_mfma_f32_16x16x32_bf16 with fp32 accumulation: each bf16 x bf16 product is exact, the dropped terms are O(2^-24 |a||b|), so the
// result has the error class of the fp32 fma chain (measured against fp64: 1.0-1.4e-7 sum|a.b| at K = 300, the exact kernel 1.4-1.5e-7;
// tools/micro/x3_rows.hip) at 6 x 16 cycles per 32 k instead of 8 x 32.
//
//  * x3_rows_kernel: A . B / A . B^T on whole rows of A (the structure of gemm_rows_kernel).  64 rows per block, taken in K CHUNKS of
//    KC = 160 (or 128) columns: three bf16 planes of 64 x KC live in LDS (63 KB: two blocks per CU), the accumulators live across the
//    chunks.  Every wave multiplies all 64 rows by its own 80 columns per pass; B fragments -- the three planes of a (column tile,
//    k-step) are 3 KB of consecutive bytes in a prep kernel's fragment order -- come straight from L2, one k-step ahead, in a ring of
//    five slots refilled tile by tile.  Two N segments (the dual launch), two K segments (the k-concatenated one) and the GATE / POST
//    epilogues are the ones of gemm_rows_kernel, arithmetic and order unchanged.
//  * x3_tn_kernel: A^T . B over the node dimension (the weight gradients).  Both operands are k-strided fp32; a thread loads an
//    8 (k) x 4 (columns) patch, splits it and writes three k-contiguous 16-byte pieces per column: the transpose happens in registers and
//    the LDS image [plane][row][32 k] is the MFMA's fragment layout.  One 8-wave block per CU, tile 160 x 320, ONE image of three planes
//    (115 KB), split-K slabs combined in slab order by splitk_reduce_kernel (deterministic).
// Both run at the socket's power cap (1.4 kW, shader clock ~1.97 GHz: profiles/r05_x3_rows_clocks.txt): what they save is energy per
// product as much as pipe time.
#include "common.h"
#include "gemm_call.h"

#include <algorithm>

namespace geogcn {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int TPB = 256;
constexpr uint32_t kOob = 0x80000000u;

__device__ __forceinline__ uint32_t bf16_pack(float lo, float hi) {       // v_cvt_pk_bf16_f32: round to nearest even, two at a time
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
// two fp32 values -> their three bf16 terms, packed pairwise (low half = x0's term); the residuals are exact in fp32
__device__ __forceinline__ void split3_pair(float x0, float x1, uint32_t (&p)[3]) {
    p[0] = bf16_pack(x0, x1);
    const float r0 = x0 - __uint_as_float(p[0] << 16), r1 = x1 - __uint_as_float(p[0] & 0xffff0000u);
    p[1] = bf16_pack(r0, r1);
    const float s0 = r0 - __uint_as_float(p[1] << 16), s1 = r1 - __uint_as_float(p[1] & 0xffff0000u);
    p[2] = bf16_pack(s0, s1);
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t mk_rsrc(const void* base, int64_t bytes) {
    // (base and bytes are wave-uniform; readfirstlane states it, or the descriptor is loaded through a per-lane waterfall loop)
    const uint32_t n = bytes <= 0 ? 0u : (bytes > 0x7FFFFFFFll ? 0x7FFFFFFFu : (uint32_t)bytes);
    const uint64_t b = reinterpret_cast<uint64_t>(base);
    const uint64_t bu = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(b >> 32)) << 32) |
                        (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)b);
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(bu), 0, __builtin_amdgcn_readfirstlane(n), 0x00020000);
}

// the six largest cross terms of one 32-k step onto MR accumulators, smallest first; the row tiles are interleaved so that no MFMA
// waits for the one before it.  Operands swapped (B fragment first): a lane owns 4 CONSECUTIVE COLUMNS of one row of C.
#define GEOGCN_X3_TERM(ACC, BF, AF, PB, PA, MR_) \
    _Pragma("unroll") for (int i_ = 0; i_ < MR_; ++i_) ACC(i_) = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BF[PB], AF(i_)[PA], ACC(i_), 0, 0, 0);
#define GEOGCN_X3_SIX(ACC, BF, AF, MR_)                                                                                   \
    GEOGCN_X3_TERM(ACC, BF, AF, 0, 2, MR_) GEOGCN_X3_TERM(ACC, BF, AF, 2, 0, MR_) GEOGCN_X3_TERM(ACC, BF, AF, 1, 1, MR_) \
    GEOGCN_X3_TERM(ACC, BF, AF, 0, 1, MR_) GEOGCN_X3_TERM(ACC, BF, AF, 1, 0, MR_) GEOGCN_X3_TERM(ACC, BF, AF, 0, 0, MR_)

// ---- weights -> fragment order, three planes: [column tile][k-step][plane][lane][8 bf16] -------------------------------------------------
// One array serves a whole launch: N segment q's tiles start at tile_base, K segment q's k-steps at kstep_base (every tile row has
// nk_total k-steps).  Entries beyond K / N are zero.
__global__ __launch_bounds__(TPB) void x3_prep_b_kernel(const float* __restrict__ W, int64_t ldw, int K, int N, int b_is_nk, int n_tiles,
                                                        int nk, int tile_base, int kstep_base, int nk_total,
                                                        unsigned short* __restrict__ out) {
    const int64_t total = (int64_t)n_tiles * nk * 512;
    for (int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x; e < total; e += (int64_t)gridDim.x * TPB) {
        const int el = (int)(e & 7), lane = (int)((e >> 3) & 63);
        const int64_t f = e >> 9;
        const int kt = (int)(f % nk), nt = (int)(f / nk);
        const int n = nt * 16 + (lane & 15), k = kt * 32 + (lane >> 4) * 8 + el;
        float x = 0.f;
        if (k < K && n < N) x = b_is_nk ? W[(int64_t)n * ldw + k] : W[(int64_t)k * ldw + n];
        uint32_t p[3];
        split3_pair(x, 0.f, p);
        unsigned short* o = out + (((int64_t)(tile_base + nt) * nk_total + kstep_base + kt) * 3) * 512 + lane * 8 + el;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) o[pl * 512] = (unsigned short)(p[pl] & 0xffffu);
    }
}

constexpr int kBM = 64, kWCT = 5;

struct X3RowsArgs {
    int64_t M;
    int n_mt;
    int n_kseg;
    const float* A[2]; int64_t lda[2]; int K[2];
    int nch[2];                         // chunks of each K segment
    const unsigned short* Bf; int nk_total;
    int n_nseg; int passes[2]; int wct[2]; int tile_base[2];
    float* C[2]; int64_t ldc[2]; const float* bias[2]; int64_t N[2]; int act_on[2];
    int accumulate;
    const float* gateG; int64_t ldg; const float* gateT; int64_t ldt;                              // as RowsArgs (gemm.hip)
    const float* postY; int64_t ldy; const uint8_t* postKeep; int64_t postF; float postScale;
    int panel_w; int64_t panel_R;       // single products only: C[0] as feature panels [N / panel_w][panel_R][panel_w] (the all-to-all's send layout, gemm.hip)
};

template <int KC, int ACT, bool GATE = false, bool POST = false>
__global__ __launch_bounds__(TPB, 2) void x3_rows_kernel(const X3RowsArgs a) {
    constexpr int BM = kBM, WCT = kWCT, MR = BM / 16;
    constexpr int PITCH = KC * 2 + 16;           // bytes per LDS row and plane: an odd multiple of 16 -> conflict-free ds_read_b128
    constexpr int PLANE = BM * PITCH;
    constexpr int F4R = KC / 4;
    constexpr int ITERS = BM * F4R / TPB;
    constexpr int KS = KC / 32;                  // k-steps per chunk
    static_assert(BM * F4R % TPB == 0 && (PITCH / 16) % 2 == 1, "chunk shape");
    extern __shared__ __attribute__((aligned(16))) unsigned char As[];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int P = a.n_nseg == 2 ? a.passes[0] + a.passes[1] : a.passes[0];
    const int n_chunks = a.n_kseg == 2 ? a.nch[0] + a.nch[1] : a.nch[0];
    const int NK = a.nk_total;
    for (int mt = blockIdx.x; mt < a.n_mt; mt += gridDim.x) {
        const int64_t m0 = (int64_t)mt * BM;
        const int64_t rows_here = std::min<int64_t>(BM, a.M - m0);
#pragma unroll 1
        for (int ps = 0; ps < P; ++ps) {
            const int seg = ps >= a.passes[0] ? 1 : 0;                  // wave-uniform
            const int lps = seg ? ps - a.passes[0] : ps;
            // column group of this wave, rotated with the row tile (gemm_rows_kernel: the group whose last tile is all padding visits every SIMD)
            const int cg = (wid + mt) & 3;
            const int w = seg ? a.wct[1] : a.wct[0];
            const int ltile0 = (cg * (seg ? a.passes[1] : a.passes[0]) + lps) * w;      // first column tile inside the segment
            const int tile0 = (seg ? a.tile_base[1] : a.tile_base[0]) + ltile0;          // ... inside the fragment array
            const bool last_real = __builtin_amdgcn_readfirstlane((int)(w == WCT && (int64_t)(ltile0 + WCT - 1) * 16 < (seg ? a.N[1] : a.N[0]))) != 0;
            const int n_tiles_all = (a.n_nseg == 2 ? a.tile_base[1] + 4 * a.passes[1] * WCT : 4 * a.passes[0] * WCT);
            const __amdgpu_buffer_rsrc_t brs = mk_rsrc(a.Bf, (int64_t)n_tiles_all * NK * 3072);
            f32x4 acc[MR][WCT];
#pragma unroll
            for (int i = 0; i < MR; ++i)
#pragma unroll
                for (int j = 0; j < WCT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            // B ring: slot j holds the three planes of column tile j for the k-step about to be multiplied and is refilled with the next
            // k-step's as soon as its MFMAs have been issued; nothing is requested past the last step (a wave-uniform branch: the re-read
            // of the last step that a branch-free form costs was a tenth of the B stream: -0.05 ms per step, profiles/
            // r05_x3_no_wasted_request_ab.txt).  Carrying the ring across passes and row tiles -- the next pass's first fragments
            // requested at the last k-step, ahead of the epilogue's stores -- is 1.6 % faster still in the micro-benchmark
            // (r05_x3_rows_micro_v18.txt) and 13 % SLOWER here: with this kernel's epilogues the loop-carried ring costs 50-110
            // spilled registers (r05_x3_carried_ring_ab.txt); not taken
            bf16x8 ring[WCT][3];
            auto bload = [&](bf16x8 (&b)[3], int j, int kt) {
                if (kt >= NK) return;
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    b[pl] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(brs, lane * 16, (((tile0 + j) * NK + kt) * 3 + pl) * 1024, 0));
            };
#pragma unroll
            for (int j = 0; j < WCT - 1; ++j) bload(ring[j], j, 0);
            if (last_real) bload(ring[WCT - 1], WCT - 1, 0);
#pragma unroll 1
            for (int c = 0; c < n_chunks; ++c) {
                {
                    const int ks = c >= a.nch[0] ? 1 : 0;           // (n_kseg == 1: nch[0] = n_chunks)
                    const int ch = ks ? c - a.nch[0] : c;
                    // (an opaque copy of the thread index: the per-thread offsets become per-chunk work instead of ~40 live registers)
                    int tt = tid;
                    asm volatile("" : "+v"(tt));
                    const float* Ap = ks ? a.A[1] : a.A[0];
                    const int64_t lda = ks ? a.lda[1] : a.lda[0];
                    const int K4 = ((ks ? a.K[1] : a.K[0]) + 3) & ~3;       // pad columns up to roundup4(K) are zero; beyond: not read
                    const __amdgpu_buffer_rsrc_t rs = mk_rsrc(Ap + m0 * lda, rows_here * lda * 4);
                    const uint32_t ld4 = (uint32_t)lda * 4u;
                    f32x4 v[ITERS];
#pragma unroll
                    for (int i = 0; i < ITERS; ++i) {
                        const int idx = tt + TPB * i;
                        const int r = idx / F4R, cc = idx - r * F4R;
                        const int kcol = ch * KC + cc * 4;
                        v[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(kcol < K4 ? (uint32_t)r * ld4 + (uint32_t)kcol * 4u : kOob), 0, 0));
                    }
                    __syncthreads();            // everybody done with the chunk before
#pragma unroll
                    for (int i = 0; i < ITERS; ++i) {
                        const int idx = tt + TPB * i;
                        const int r = idx / F4R, cc = idx - r * F4R;
                        uint32_t p01[3], p23[3];
                        split3_pair(v[i][0], v[i][1], p01);
                        split3_pair(v[i][2], v[i][3], p23);
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<uint2*>(As + pl * PLANE + r * PITCH + cc * 8) = make_uint2(p01[pl], p23[pl]);
                    }
                }
                __syncthreads();
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const int kt = c * KS + s;
                    bf16x8 af[MR][3];
#pragma unroll
                    for (int i = 0; i < MR; ++i)
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl)
                            af[i][pl] = *reinterpret_cast<const bf16x8*>(As + pl * PLANE + (i * 16 + li) * PITCH + s * 64 + lg * 16);
#define GEOGCN_ACC_(i_) acc[i_][j]
#define GEOGCN_AF_(i_) af[i_]
#pragma unroll
                    for (int j = 0; j < WCT - 1; ++j) {
                        GEOGCN_X3_SIX(GEOGCN_ACC_, ring[j], GEOGCN_AF_, MR)
                        bload(ring[j], j, kt + 1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (last_real) {          // (an all-padding tile would only multiply zeros: its accumulators stay 0, its stores are masked)
                        constexpr int j = WCT - 1;
                        GEOGCN_X3_SIX(GEOGCN_ACC_, ring[j], GEOGCN_AF_, MR)
                        bload(ring[j], j, kt + 1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
#undef GEOGCN_ACC_
#undef GEOGCN_AF_
                }
            }
            // ---- epilogue: the arithmetic of gemm_rows_kernel (gemm.hip), in its order ----
            float* Cout = seg ? a.C[1] : a.C[0];
            const int64_t ldc = seg ? a.ldc[1] : a.ldc[0];
            const float* bias = seg ? a.bias[1] : a.bias[0];
            // (columns from ltile0 + w on belong to the next wave)
            const int64_t Nseg = std::min<int64_t>(seg ? a.N[1] : a.N[0], (int64_t)(ltile0 + w) * 16);
            const bool act_on = (seg ? a.act_on[1] : a.act_on[0]) != 0;
            const int64_t ncol0 = (int64_t)ltile0 * 16;
            float bcol[WCT][4];
#pragma unroll
            for (int j = 0; j < WCT; ++j) {
                const int64_t col0 = ncol0 + j * 16 + lg * 4;
#pragma unroll
                for (int r = 0; r < 4; ++r) bcol[j][r] = (bias && col0 + r < Nseg) ? bias[col0 + r] : 0.f;
            }
            const float* const gG = GATE ? a.gateG : nullptr; const float* const gT = GATE ? a.gateT : nullptr;
            const int64_t ldg = a.ldg, ldt = a.ldt, Mrows = a.M;
            const bool accum = a.accumulate != 0;
            const bool extra = accum || gG != nullptr;
            f32x4 ea[WCT], eb[WCT];
            const int64_t lda_e = gG ? ldg : ldc;
            const __amdgpu_buffer_rsrc_t ersA = mk_rsrc((gG ? gG + m0 * ldg : Cout + m0 * ldc), extra ? rows_here * lda_e * 4 : 0);
            const __amdgpu_buffer_rsrc_t ersB = mk_rsrc(gG ? gT + m0 * ldt : Cout, gG ? rows_here * ldt * 4 : 0);
            const uint32_t lda_e4 = (uint32_t)lda_e * 4u, ldt4 = (uint32_t)ldt * 4u;
            const __amdgpu_buffer_rsrc_t prsY = mk_rsrc(POST ? a.postY + m0 * a.ldy : Cout, POST ? rows_here * a.ldy * 4 : 0);
            const __amdgpu_buffer_rsrc_t prsK = mk_rsrc(POST ? reinterpret_cast<const float*>(a.postKeep + m0 * a.postF) : Cout,
                                                        POST ? rows_here * a.postF : 0);
            const uint32_t ldy4 = POST ? (uint32_t)a.ldy * 4u : 0u, pF = POST ? (uint32_t)a.postF : 0u;
            const float pscale = POST ? a.postScale : 0.f;
            auto eload = [&](int i, int j) __attribute__((always_inline)) {
                const int64_t col0 = ncol0 + j * 16 + lg * 4;
                const uint32_t r = (uint32_t)(i * 16 + li), cb = (uint32_t)col0 * 4u;
                const bool ok = col0 < Nseg;
                ea[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ersA, (int)(ok ? r * lda_e4 + cb : kOob), 0, 0));
                if constexpr (GATE && !POST) eb[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ersB, (int)(ok ? r * ldt4 + cb : kOob), 0, 0));
            };
            if (extra) {
#pragma unroll
                for (int j = 0; j < WCT; ++j) eload(0, j);
            }
            f32x4 pt = f32x4{0.f, 0.f, 0.f, 0.f}, py = pt;
            uint32_t pk = 0;
            auto pload = [&](int i, int j) __attribute__((always_inline)) {
                const int64_t col0 = ncol0 + j * 16 + lg * 4;
                const uint32_t pr = (uint32_t)(i * 16 + li), pc = (uint32_t)col0;
                const bool pok = col0 < Nseg;
                pt = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ersB, (int)(pok ? pr * ldt4 + pc * 4u : kOob), 0, 0));
                py = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(prsY, (int)(pok ? pr * ldy4 + pc * 4u : kOob), 0, 0));
                pk = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(prsK, (int)(pok ? pr * pF + pc : kOob), 0, 0);
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
                    if (row_ok && col0 < Nseg) {
                        float* dst = crow + col0;
                        if constexpr (!GATE && !POST) {
                            if (a.panel_w) {          // (panel_w % 4 == 0: the four columns of a lane lie in one panel)
                                const int64_t q = col0 / a.panel_w;
                                dst = Cout + (q * a.panel_R + row) * a.panel_w + (col0 - q * a.panel_w);
                            }
                        }
                        *reinterpret_cast<float4*>(dst) = make_float4(x[0], x[1], x[2], x[3]);
                    }
                    if (extra && i + 1 < MR) eload(i + 1, j);
                    if constexpr (POST) {
                        if (j + 1 < WCT) pload(i, j + 1);
                        else if (i + 1 < MR) pload(i + 1, 0);
                    }
                }
            }
        }
    }
}

// ---- A^T . B ----------------------------------------------------------------------------------------------------------------------------
constexpr int BKH = 32, ROWB = 80;      // k per stage = one MFMA depth; bytes per LDS row: 32 bf16 + 16 B pad (an odd multiple of 16)

template <int BM, int BN>
__global__ __launch_bounds__(512, 1) void x3_tn_kernel(const X3TnCall a) {
    constexpr int NTH = 512, kASplit = 192;                  // threads [0,192): A patches, [192,512): B patches
    constexpr int MR = BM / 32, NR = BN / 64;                // 2 x 4 waves, wave tile (BM/2) x (BN/4)
    constexpr int kAItems = 4 * (BM / 4), kBItems = 4 * (BN / 4);
    static_assert(kAItems <= kASplit && kBItems <= NTH - kASplit, "patch lists must fit the thread ranges");
    constexpr int kImgA = BM * ROWB, kPlane = (BM + BN) * ROWB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 2, wn = wid & 3, li = lane & 15, lg = lane >> 4;

    // Block b runs on XCD b % 8: all tiles of one K slab go to the SAME XCD and start together (they read the same rows of A and B: one
    // fetches a row from HBM, the others hit that XCD's L2).  Slabs beyond nsplit (grid rounded up to whole XCD rounds) do nothing.
    const int b = blockIdx.x;
    const int xcd = b % kNumXCD, s = b / kNumXCD;
    const int tiles = a.n_nt * a.n_mt;
    const int tile = s % tiles;
    const int ntile = __builtin_amdgcn_readfirstlane(tile % a.n_nt);
    const int mt = __builtin_amdgcn_readfirstlane(tile / a.n_nt);
    const int z = __builtin_amdgcn_readfirstlane(xcd + kNumXCD * (s / tiles));
    if (z >= a.nsplit) return;
    const int seg = ntile / a.nt_per_seg, nt = ntile % a.nt_per_seg;
    const int64_t M = a.M, N = seg ? a.N[1] : a.N[0];
    const int64_t m0 = (int64_t)mt * BM, n0 = (int64_t)nt * BN;
    const int64_t kbeg = (int64_t)z * a.kchunk, kend = std::min<int64_t>(a.K, kbeg + a.kchunk);
    const int nk = (int)((kend - kbeg + BKH - 1) / BKH);

    // my patch: operand, k8 group (8 reduction rows), c4 group (4 columns)
    const bool isA = tid < kASplit;
    const int it = isA ? tid : tid - kASplit;
    const int cols4 = isA ? BM / 4 : BN / 4;
    const bool active = it < 4 * cols4;
    const int k8 = it / cols4, c4 = it % cols4;
    const float* Pp = isA ? a.A : (seg ? a.B[1] : a.B[0]);
    const int64_t ld = isA ? a.lda : (seg ? a.ldb[1] : a.ldb[0]);
    const int64_t c0 = isA ? m0 : n0;
    const int64_t ctot = isA ? M : N;
    // (a float4 is wholly inside [0, roundup4(columns)) or outside: pad columns are zero by the geogcn.h convention)
    const bool col_ok = active && (c4 * 4 < ((ctot + 3) & ~(int64_t)3) - c0);
    f32x4 patch[8];
    auto gload = [&](int kt) {
        const int64_t k0 = kbeg + (int64_t)kt * BKH;
        // the descriptor ends with the slab: rows past it (and stages past the slab) read as zeros in hardware
        const __amdgpu_buffer_rsrc_t rs = mk_rsrc(Pp + k0 * ld + c0, ((kend - k0) * ld - c0) * 4);
        const uint32_t ld4 = (uint32_t)ld * 4u;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            patch[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(col_ok ? (uint32_t)(k8 * 8 + i) * ld4 + (uint32_t)c4 * 16u : kOob), 0, 0));
    };
    // registers -> the three planes of the LDS image: column e of the patch = component e of every row
    auto sstore = [&]() {
        if (!active) return;
        unsigned char* img = smem_raw + (isA ? 0 : kImgA);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            uint32_t p[4][3];
#pragma unroll
            for (int q = 0; q < 4; ++q) split3_pair(patch[2 * q][e], patch[2 * q + 1][e], p[q]);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                *reinterpret_cast<uint4*>(img + pl * kPlane + (c4 * 4 + e) * ROWB + k8 * 16) = make_uint4(p[0][pl], p[1][pl], p[2][pl], p[3][pl]);
        }
    };

    f32x4 acc[MR][NR];
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    gload(0);
#pragma unroll 1
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();                    // everybody done reading the stage before
        sstore();
        gload(kt + 1);                      // in flight during this stage's MFMAs
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        const unsigned char* Asm = smem_raw;
        const unsigned char* Bsm = smem_raw + kImgA;
        bf16x8 af[MR][3];
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                af[i][pl] = *reinterpret_cast<const bf16x8*>(Asm + pl * kPlane + (wm * (BM / 2) + i * 16 + li) * ROWB + lg * 16);
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            bf16x8 bf[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                bf[pl] = *reinterpret_cast<const bf16x8*>(Bsm + pl * kPlane + (wn * (BN / 4) + j * 16 + li) * ROWB + lg * 16);
#define GEOGCN_ACC_(i_) acc[i_][j]
#define GEOGCN_AF_(i_) af[i_]
            GEOGCN_X3_SIX(GEOGCN_ACC_, bf, GEOGCN_AF_, MR)
#undef GEOGCN_ACC_
#undef GEOGCN_AF_
        }
    }
    // slab z: lane (li, lg) holds C[row = li][col = 4 lg + r] of each 16 x 16 sub-tile
    float* Wz = a.W + (int64_t)z * M * a.ldw + (seg ? a.seg_w : 0);
    const int64_t n_store = (N + 3) & ~(int64_t)3;
#pragma unroll
    for (int i = 0; i < MR; ++i) {
        const int64_t row = m0 + wm * (BM / 2) + i * 16 + li;
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const int64_t col0 = n0 + wn * (BN / 4) + j * 16 + lg * 4;
            f32x4 x = acc[i][j];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (col0 + q >= N) x[q] = 0.f;
            if (row < M && col0 < n_store) *reinterpret_cast<f32x4*>(Wz + row * a.ldw + col0) = x;
        }
    }
}

// ---- whole-rows kernel: which calls take it, its workspace, its launch -----------------------------------------------------------------
// Below this many rows an A . B stays on the exact fp32 kernels (the weights' split + fragment-order pass is a launch of its own and
// 64-row blocks no longer fill the chip).  The test seam GEOGCN_X3_ROWS_MIN_M (common.h) lowers it -- and lifts the rule about
// padded columns below -- so that the model-level oracle tests at CMU / fixture sizes (12-wide layers included) run THIS kernel and not
// the exact one under the label 'bf16x3'.
constexpr int64_t kX3RowsMinM = 4096;          // (32,768 until round 6; the CMU-shape step, 9,475 rows: 1.118 -> 1.010 ms with its products on this kernel, profiles/r06_cmu_x3_threshold.txt)
inline int64_t x3_rows_min_m() { return test_seam_i64("GEOGCN_X3_ROWS_MIN_M", kX3RowsMinM); }
constexpr int64_t kX3RowsMaxN = 1024, kX3RowsMaxK = 1024;          // (round 6: 640 until then; the reference's WORLD run is 900 / 930 wide)
inline int rows_passes(int64_t N) { return (int)cdiv(N, 4 * kWCT * 16); }                  // column passes of 320 (300 -> 1, 600 -> 2, 900 -> 3)
inline int rows_wct(int64_t N) { return (int)cdiv(cdiv(N, 16), 4 * rows_passes(N)); }      // 5, or 4 (N <= 256, 321..512, 961..1024), or fewer
inline int chunks_of(int64_t K, int kc) { return (int)cdiv(K, kc); }
inline size_t seg_tiles(int64_t N) { return (size_t)4 * rows_passes(N) * kWCT; }

template <int KC>
int launch_x3_rows(const X3RowsArgs& a, int act, hipStream_t st) {
    constexpr int lds = 3 * kBM * (KC * 2 + 16);
    const int G = (int)std::min<int64_t>((int64_t)kNumCU * 2, a.n_mt);
#define GEOGCN_XR(...)                                                                                           \
    do {                                                                                                         \
        auto kern = x3_rows_kernel<KC, __VA_ARGS__>;                                                             \
        static LdsAttrOnce lds_once;                                                                           \
        if (const int rc_ = lds_once.ensure((const void*)kern, (int)(lds))) return rc_; \
        hipLaunchKernelGGL(kern, dim3((unsigned)G), dim3(TPB), lds, st, a);                                      \
        GEOGCN_LAUNCH_CHECK("x3_rows_kernel");                                                                   \
    } while (0)
    if (a.gateG && a.postY) GEOGCN_XR(GEOGCN_ACT_NONE, true, true);
    else if (a.gateG) GEOGCN_XR(GEOGCN_ACT_NONE, true);
    else if (act == GEOGCN_ACT_TANH) GEOGCN_XR(GEOGCN_ACT_TANH);
    else if (act == GEOGCN_ACT_SIGMOID) GEOGCN_XR(GEOGCN_ACT_SIGMOID);
    else GEOGCN_XR(GEOGCN_ACT_NONE);
#undef GEOGCN_XR
    return 0;
}

}  // namespace

// Shapes: every call gemm.hip's whole-rows kernel takes (the fused launches, single A . B^T, single A . B of 256 / 512 columns) and,
// here, single A . B of ANY width up to 1,024 (the staged bf16x3 kernel is what such a call would run on otherwise); K up to 1,024 in
// chunks of 160 (K <= 256: 128).  Returns the chunk width, 0 = not taken.
int x3_rows_kc(const GemmCall& c, bool transA, bool transB) {
    (void)transB;
#ifdef GEOGCN_NO_X3_ROWS          // A/B build only (GEOGCN_BUILD_DEFINES)
    return 0;
#endif
    const int64_t min_m = x3_rows_min_m();
    if (transA || c.M < min_m || c.precision != GEOGCN_GEMM_BF16X3) return 0;
    // (round 6) panel outputs -- the partitioned path's H . W written straight into the all-to-all's send layout -- for single products
    if (c.panel_w && (c.n_nseg != 1 || c.n_kseg != 1 || c.accumulate || c.gateG || c.postY)) return 0;
    const bool seam = min_m != kX3RowsMinM;                  // under the test seam: every shape the kernel CAN compute, not only those it is fast on
    const int64_t kmax = c.n_kseg == 2 ? std::max(c.K[0], c.K[1]) : c.K[0];
    if (kmax > kX3RowsMaxK) return 0;
    const int kc = cdiv(kmax, 32) * 32 <= 256 ? 128 : 160;
    // (two K segments over several column passes -- dH = dZ . Wh^T + dU . Wt^T at 600 / 900 columns -- since round 6: a pass walks the chunks
    //  of both segments into its accumulators like a single product's; gemm.hip's exact whole-rows kernel keeps one pass)
    for (int q = 0; q < c.n_nseg; ++q) {
        if (c.N[q] > kX3RowsMaxN) return 0;
        const int64_t cols = (int64_t)rows_passes(c.N[q]) * 4 * std::max(rows_wct(c.N[q]), 4) * 16;
        if (!seam && (cols - c.N[q]) * 4 > cols) return 0;    // at most a quarter of a pass multiplies zero columns
    }
    return kc;
}

size_t x3_rows_ws_bytes(const GemmCall& c, int kc) {
    const int ks = kc / 32;
    const int nk_total = (c.n_kseg == 2 ? chunks_of(c.K[0], kc) + chunks_of(c.K[1], kc) : chunks_of(c.K[0], kc)) * ks;
    const size_t tiles = seg_tiles(c.N[0]) + (c.n_nseg == 2 ? seg_tiles(c.N[1]) : 0);
    return tiles * (size_t)nk_total * 3072;
}

int x3_run_rows(int kc, bool transB, const GemmCall& c, void* ws, hipStream_t st) {
    X3RowsArgs a{};
    a.M = c.M;
    a.n_mt = (int)cdiv(c.M, kBM);
    a.n_kseg = c.n_kseg;
    a.n_nseg = c.n_nseg;
    a.accumulate = c.accumulate;
    a.gateG = c.gateG; a.ldg = c.ldg; a.gateT = c.gateT; a.ldt = c.ldt;
    a.postY = c.postY; a.ldy = c.ldy; a.postKeep = c.postKeep; a.postF = c.postF; a.postScale = c.postScale;
    a.panel_w = c.panel_w; a.panel_R = c.panel_R;
    const int ks = kc / 32;
    for (int q = 0; q < 2; ++q) {
        a.A[q] = c.A[q]; a.lda[q] = c.lda[q]; a.K[q] = (int)c.K[q];
        a.nch[q] = (q == 0 || c.n_kseg == 2) ? chunks_of(c.K[q], kc) : 0;
        a.C[q] = c.C[q]; a.ldc[q] = c.ldc[q]; a.bias[q] = c.bias[q]; a.N[q] = c.N[q];
        a.act_on[q] = c.act[q] != GEOGCN_ACT_NONE;
        a.passes[q] = c.N[q] > 0 ? rows_passes(c.N[q]) : 0;
        a.wct[q] = c.N[q] > 0 ? std::max(rows_wct(c.N[q]), 4) : kWCT;
    }
    a.tile_base[0] = 0;
    a.tile_base[1] = c.n_nseg == 2 ? (int)seg_tiles(c.N[0]) : 0;
    a.nk_total = (a.nch[0] + (c.n_kseg == 2 ? a.nch[1] : 0)) * ks;
    a.Bf = (const unsigned short*)ws;
    // weights -> fragment order: slot q = N segment | K segment
    const int n_slots = (c.n_nseg == 2 || c.n_kseg == 2) ? 2 : 1;
    for (int q = 0; q < n_slots; ++q) {
        const int64_t N = c.n_kseg == 2 ? c.N[0] : c.N[q];
        const int64_t K = c.n_kseg == 2 ? c.K[q] : c.K[0];
        const int n_tiles = (int)seg_tiles(N);
        const int nk = (c.n_kseg == 2 ? a.nch[q] : a.nch[0]) * ks;
        const int tile_base = c.n_nseg == 2 ? a.tile_base[q] : 0;
        const int kstep_base = (c.n_kseg == 2 && q == 1) ? a.nch[0] * ks : 0;
        const unsigned grid = (unsigned)std::min<int64_t>(cdiv((int64_t)n_tiles * nk * 512, TPB), 1024);
        hipLaunchKernelGGL(x3_prep_b_kernel, dim3(grid), dim3(TPB), 0, st, c.B[q], c.ldb[q], (int)K, (int)N, transB ? 1 : 0, n_tiles, nk,
                           tile_base, kstep_base, a.nk_total, (unsigned short*)ws);
        GEOGCN_LAUNCH_CHECK("x3_prep_b_kernel");
    }
    const int act = c.act[0] != GEOGCN_ACT_NONE ? c.act[0] : c.act[1];
    if (kc == 160) return launch_x3_rows<160>(a, act, st);
    return launch_x3_rows<128>(a, act, st);
}

bool x3_tn_takes(int bm, int bn) {
#ifdef GEOGCN_NO_X3_TN            // A/B build only
    return false;
#endif
    return (bm == 160 || bm == 128) && (bn == 320 || bn == 256);
}

int x3_tn_launch(int bm, int bn, const X3TnCall& t, hipStream_t st) {
    const int T = t.n_mt * t.n_nt;
    const dim3 grid((unsigned)(cdiv(t.nsplit, kNumXCD) * kNumXCD * T));
#define GEOGCN_XT(BM_, BN_)                                                                                      \
    do {                                                                                                         \
        auto kern = x3_tn_kernel<BM_, BN_>;                                                                      \
        constexpr int lds = 3 * (BM_ + BN_) * ROWB;                                                              \
        static LdsAttrOnce lds_once;                                                                           \
        if (const int rc_ = lds_once.ensure((const void*)kern, (int)(lds))) return rc_; \
        hipLaunchKernelGGL(kern, grid, dim3(512), lds, st, t);                                                   \
        GEOGCN_LAUNCH_CHECK("x3_tn_kernel");                                                                     \
    } while (0)
    if (bm == 160 && bn == 320) GEOGCN_XT(160, 320);
    else if (bm == 160 && bn == 256) GEOGCN_XT(160, 256);
    else if (bm == 128 && bn == 320) GEOGCN_XT(128, 320);
    else if (bm == 128 && bn == 256) GEOGCN_XT(128, 256);
    else {
        set_error("x3_tn_launch: no kernel for tile %dx%d", bm, bn);
        return GEOGCN_E_ARG;
    }
#undef GEOGCN_XT
    return 0;
}

}  // namespace geogcn
