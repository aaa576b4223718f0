// Micro-benchmark (NOT part of the product): the whole-rows GEMM with the fp32-class split-bf16 ("bf16x3") contraction.
//   C[M x N] = A[M x K] . W,  M = 440,000, K = 300, N = 300 (one product) or 600 (the highway block's dual launch)
// Every fp32 value is split EXACTLY into three bf16 terms (a = a1 + a2 + a3) and the product is formed from the six largest
// cross terms on v_mfma_f32_16x16x32_bf16 (fp32 accumulate): 6 bf16 MFMAs of 16 cycles per 32 k against 8 fp32 MFMAs of 32.
// Structure: 64 rows of A per block; the rows are taken in K CHUNKS of KC = 160 (three bf16 planes of 64 x 160 = 63 KB of LDS:
// two blocks per CU), the accumulators live across the chunks; B fragments (three planes, fragment order) straight from L2,
// one k-step ahead in a ring of WCT slots refilled tile by tile.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/x3_rows.hip -o tools/micro/bin/x3_rows && tools/micro/bin/x3_rows
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e_ = (x);                                                   \
        if (e_ != hipSuccess) {                                                \
            printf("%s: %s\n", #x, hipGetErrorString(e_));                     \
            exit(1);                                                           \
        }                                                                      \
    } while (0)

__device__ __forceinline__ uint32_t bf16_pack(float lo, float hi) {
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
// two fp32 values -> their three bf16 terms, packed pairwise (lo half = x0's term); residuals are exact in fp32
__device__ __forceinline__ void split3_pair(float x0, float x1, uint32_t (&p)[3]) {
    p[0] = bf16_pack(x0, x1);
    const float r0 = x0 - __uint_as_float(p[0] << 16), r1 = x1 - __uint_as_float(p[0] & 0xffff0000u);
    p[1] = bf16_pack(r0, r1);
    const float s0 = r0 - __uint_as_float(p[1] << 16), s1 = r1 - __uint_as_float(p[1] & 0xffff0000u);
    p[2] = bf16_pack(s0, s1);
}

// truncation split (probe): x = t1 + t2 + t3 EXACTLY as well (the top 8 significant bits three times), bit masks + subtracts only
__device__ __forceinline__ void split3_pair_trunc(float x0, float x1, uint32_t (&p)[3]) {
    const uint32_t u0 = __float_as_uint(x0) & 0xffff0000u, u1 = __float_as_uint(x1) & 0xffff0000u;
    p[0] = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
    const float r0 = x0 - __uint_as_float(u0), r1 = x1 - __uint_as_float(u1);
    const uint32_t v0 = __float_as_uint(r0) & 0xffff0000u, v1 = __float_as_uint(r1) & 0xffff0000u;
    p[1] = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
    const float s0 = r0 - __uint_as_float(v0), s1 = r1 - __uint_as_float(v1);
    p[2] = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t mk_rsrc(const void* base, int64_t bytes) {
    const uint32_t n = bytes <= 0 ? 0u : (bytes > 0x7FFFFFFFll ? 0x7FFFFFFFu : (uint32_t)bytes);
    const uint64_t b = reinterpret_cast<uint64_t>(base);
    const uint64_t bu = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(b >> 32)) << 32) |
                        (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)b);
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(bu), 0, __builtin_amdgcn_readfirstlane(n), 0x00020000);
}

// PROBE: 1 = no C stores, 2 = no A loads (LDS holds garbage), 4 = no MFMAs, 8 = no B loads, 16 = one B plane loaded (a third of the
// B bytes), 32 = every B load from the first 15 KB (L1 hits), 64 = no LDS fragment reads,
// 256 = C stores as whole 128-byte lines (address pattern only: wrong values), 512 = non-temporal C stores, 1024 = non-temporal A loads
template <int KC, int NCH, int WCT, int PF, int PROBE = 0, int BM = 64, int RD = 1, int STG = 0, int CONT = 0, int AFPF = 0>
__global__ __launch_bounds__(256, (BM == 64 && RD == 1) ? 2 : 1) void x3_rows_kernel(const float* __restrict__ A, int64_t lda, int64_t M, int K,
                                                         const unsigned short* __restrict__ Bf, int N, float* __restrict__ C,
                                                         int64_t ldc, int n_mt, int passes, long long* tim = nullptr) {
    constexpr int MR = BM / 16;
    long long t_load = 0, t_mfma = 0, t_epi = 0, t_mark = 0;
    auto tick = [&]() { return (PROBE & 128) ? (long long)__builtin_readcyclecounter() : 0ll; };
    const long long t_begin = tick();
    constexpr int PITCH = KC * 2 + 16;           // bytes per LDS row and plane: an odd multiple of 16
    constexpr int PLANE = BM * PITCH;
    constexpr int F4R = KC / 4;                  // float4 per row of a chunk
    constexpr int ITERS = BM * F4R / 256;
    constexpr int KS = KC / 32;                  // k-steps per chunk
    constexpr int NK = NCH * KS;                 // k-steps in all
    static_assert(BM * F4R % 256 == 0, "a chunk must divide over the block");
    extern __shared__ __attribute__((aligned(16))) unsigned char As[];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int K4 = (K + 3) & ~3;
    const uint32_t ld4 = (uint32_t)lda * 4u;
    const __amdgpu_buffer_rsrc_t brs = mk_rsrc(Bf, (int64_t)4 * passes * WCT * NK * 3072);

    f32x4 pre[PF ? ITERS : 1];
    auto aload = [&](f32x4 (&v)[ITERS], int mt, int ch) {
        int tt = tid;
        asm volatile("" : "+v"(tt));
        const int64_t m0 = (int64_t)mt * BM;
        const int64_t rows = mt < n_mt ? (M - m0 < BM ? M - m0 : BM) : 0;
        const __amdgpu_buffer_rsrc_t rs = mk_rsrc(A + (mt < n_mt ? m0 : 0) * lda, rows * lda * 4);
#pragma unroll
        for (int i = 0; i < ITERS; ++i) {
            const int idx = tt + 256 * i;
            const int r = idx / F4R, c = idx - r * F4R;
            const int kcol = ch * KC + c * 4;
            v[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(kcol < K4 ? (uint32_t)r * ld4 + (uint32_t)kcol * 4u : 0x80000000u), 0, (PROBE & 1024) ? 2 : 0));
        }
    };
    auto astore = [&](const f32x4 (&v)[ITERS]) {
        int tt = tid;
        asm volatile("" : "+v"(tt));
#pragma unroll
        for (int i = 0; i < ITERS; ++i) {
            const int idx = tt + 256 * i;
            const int r = idx / F4R, c = idx - r * F4R;
            uint32_t p01[3], p23[3];
            if (PROBE & 2048) { split3_pair_trunc(v[i][0], v[i][1], p01); split3_pair_trunc(v[i][2], v[i][3], p23); }
            else { split3_pair(v[i][0], v[i][1], p01); split3_pair(v[i][2], v[i][3], p23); }
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<uint2*>(As + pl * PLANE + r * PITCH + c * 8) = make_uint2(p01[pl], p23[pl]);
        }
    };

    // STG: the second resident block of every CU starts late by STG x ~8k cycles, so that its load / split / store phases fall into the
    // first block's MFMA phases instead of coinciding with its load phases
    if (STG && blockIdx.x >= gridDim.x / 2) {
#pragma unroll 1
        for (int i = 0; i < STG; ++i) __builtin_amdgcn_s_sleep(127);
    }
    bf16x8 ring[RD][WCT][3];
    if (false && CONT == 1 && !(PROBE & 8)) {          // the block's first pass: its first k-step's fragments (every later pass finds its own in the ring)
#pragma unroll
        for (int d = 0; d < RD; ++d)
#pragma unroll
            for (int j = 0; j < WCT; ++j)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    ring[d][j][pl] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(brs, lane * 16, (((wid * passes * WCT + j) * NK + d) * 3 + pl) * 1024, 0));
    }
    for (int mt = blockIdx.x; mt < n_mt; mt += gridDim.x) {
        const int64_t m0 = (int64_t)mt * BM;
#pragma unroll 1
        for (int ps = 0; ps < passes; ++ps) {
            const int tile0 = (wid * passes + ps) * WCT;
            // (CONT) the column tiles of the pass after this one: the next pass of the tile, or the first pass of the block's next tile
            const int tile0_next = (wid * passes + (ps + 1 < passes ? ps + 1 : 0)) * WCT;
            const int tile0_next4 = __builtin_amdgcn_readfirstlane(tile0_next);
            if (CONT == 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (debug: every carried request landed before the pass begins)
            const int ncol0 = tile0 * 16;
            f32x4 acc[MR][WCT];
#pragma unroll
            for (int i = 0; i < MR; ++i)
#pragma unroll
                for (int j = 0; j < WCT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            // B ring: slot j holds the three planes of column tile j for the k-step about to be multiplied; it is refilled with the
            // next k-step's as soon as its MFMAs have been issued
            auto bload = [&](bf16x8 (&b)[3], int j, int kt) {
                const int kk = kt < NK ? kt : (CONT ? kt - NK : NK - 1);
                const int tsel = (CONT && kt >= NK) ? (CONT == 4 ? tile0_next4 : tile0_next) : tile0;
#pragma unroll
                for (int pl = 0; pl < ((PROBE & 16) ? 1 : 3); ++pl)
                    b[pl] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(brs, lane * 16, (PROBE & 32) ? (j * 3 + pl) * 1024 : (((tsel + j) * NK + kk) * 3 + pl) * 1024, 0));
                if (PROBE & 16) { b[1] = b[0]; b[2] = b[0]; }
            };
            if (!(PROBE & 8) && ((CONT != 1 && CONT != 4 && CONT != 5) || (mt == (int)blockIdx.x && ps == 0))) {
#pragma unroll
                for (int d = 0; d < RD; ++d)
#pragma unroll
                    for (int j = 0; j < WCT; ++j) bload(ring[d][j], j, d);
            } else if (PROBE & 8) {
#pragma unroll
                for (int d = 0; d < RD; ++d)
#pragma unroll
                for (int j = 0; j < WCT; ++j)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) ring[d][j][pl] = bf16x8{1, 2, 3, 4, 5, 6, 7, 8};
            }
#pragma unroll 1
            for (int ch = 0; ch < NCH; ++ch) {
                t_mark = tick();
                if (!(PROBE & 2)) {
                    if constexpr (PF == 1) {
                        // (prefetched: the first chunk of a pass / tile was requested during the previous chunk's MFMAs)
                        if (mt == (int)blockIdx.x && ps == 0 && ch == 0) aload(pre, mt, 0);
                        __syncthreads();          // everybody done with the chunk before
                        astore(pre);
                    } else if constexpr (PF == 2) {
                        if (ch == 0) {
                            // (the first chunk of a pass was requested ahead of the previous pass's stores)
                            if (mt == (int)blockIdx.x && ps == 0) aload(pre, mt, 0);
                            __syncthreads();
                            astore(pre);
                        } else {
                            f32x4 v[ITERS];
                            aload(v, mt, ch);
                            __syncthreads();
                            astore(v);
                        }
                    } else {
                        f32x4 v[ITERS];
                        aload(v, mt, ch);
                        __syncthreads();
                        astore(v);
                    }
                }
                __syncthreads();
                { const long long t = tick(); t_load += t - t_mark; t_mark = t; }
                if constexpr (PF == 1 && !(PROBE & 2)) {
                    // next chunk in program order: (ch + 1), else the next pass's chunk 0, else the next tile's
                    const int nch = ch + 1 < NCH ? ch + 1 : 0;
                    const int nmt = (ch + 1 < NCH || ps + 1 < passes) ? mt : mt + (int)gridDim.x;
                    aload(pre, nmt, nch);
                    __builtin_amdgcn_sched_barrier(0);
                }
                // AFPF: the A fragments of k-step ks + 1 are read from LDS while k-step ks is multiplied (a single wave per SIMD has nobody
                // to hide the ds_read latency behind)
                bf16x8 afb[AFPF == 1 ? 2 : 1][MR][3];
                auto afload = [&](bf16x8 (&a)[MR][3], int ks) {
#pragma unroll
                    for (int i = 0; i < MR; ++i)
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl) {
                            if (PROBE & 64) a[i][pl] = bf16x8{(short)i, (short)pl, 3, 4, 5, 6, 7, (short)ks};
                            else a[i][pl] = *reinterpret_cast<const bf16x8*>(As + pl * PLANE + (i * 16 + li) * PITCH + ks * 64 + lg * 16);
                        }
                };
                // AFPF == 2: only the first half of the row tiles is read a k-step ahead; the second half is requested at the top of its own
                // k-step and multiplied after the first half (24 MFMAs of cover), 144 instead of 192 fragment registers
                constexpr int MH = MR / 2;
                bf16x8 aflo[2][MH][3], afhi[MH][3];
                auto afload_half = [&](bf16x8 (&a)[MH][3], int ks, int i0) {
#pragma unroll
                    for (int i = 0; i < MH; ++i)
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl)
                            a[i][pl] = *reinterpret_cast<const bf16x8*>(As + pl * PLANE + ((i0 + i) * 16 + li) * PITCH + ks * 64 + lg * 16);
                };
                if (AFPF == 1) afload(afb[0], 0);
                if (AFPF == 2) afload_half(aflo[0], 0, 0);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const int kt = ch * KS + ks;
                    if (AFPF == 2) {
                        afload_half(afhi, ks, MH);
                        if (ks + 1 < KS) afload_half(aflo[(ks + 1) & 1], ks + 1, 0);
#pragma unroll
                        for (int j = 0; j < WCT; ++j) {
#define X3_TERMH(PB, AF, I0)                                                                                    \
    _Pragma("unroll") for (int i = 0; i < MH; ++i) acc[I0 + i][j] =                                             \
        __builtin_amdgcn_mfma_f32_16x16x32_bf16(ring[ks % RD][j][PB], AF, acc[I0 + i][j], 0, 0, 0);
                            X3_TERMH(0, aflo[ks & 1][i][2], 0) X3_TERMH(2, aflo[ks & 1][i][0], 0) X3_TERMH(1, aflo[ks & 1][i][1], 0)
                            X3_TERMH(0, aflo[ks & 1][i][1], 0) X3_TERMH(1, aflo[ks & 1][i][0], 0) X3_TERMH(0, aflo[ks & 1][i][0], 0)
                            X3_TERMH(0, afhi[i][2], MH) X3_TERMH(2, afhi[i][0], MH) X3_TERMH(1, afhi[i][1], MH)
                            X3_TERMH(0, afhi[i][1], MH) X3_TERMH(1, afhi[i][0], MH) X3_TERMH(0, afhi[i][0], MH)
#undef X3_TERMH
                            if (!(PROBE & 8) && !((CONT == 2 || CONT == 5) && kt + RD >= NK)) bload(ring[ks % RD][j], j, kt + RD);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        continue;
                    }
                    if (AFPF) { if (ks + 1 < KS) afload(afb[(ks + 1) & 1], ks + 1); }
                    else afload(afb[0], ks);
                    bf16x8 (&af)[MR][3] = afb[AFPF == 1 ? (ks & 1) : 0];
#pragma unroll
                    for (int j = 0; j < WCT; ++j) {
                        if (!(PROBE & 4)) {
                            // the six largest cross terms, smallest first; the four row tiles interleaved so that no MFMA waits for
                            // the one before it (operands swapped: a lane owns 4 consecutive columns of one row of C)
#define X3_TERM(PB, PA)                                                                                         \
    _Pragma("unroll") for (int i = 0; i < MR; ++i) acc[i][j] =                                                  \
        __builtin_amdgcn_mfma_f32_16x16x32_bf16(ring[ks % RD][j][PB], af[i][PA], acc[i][j], 0, 0, 0);
                            X3_TERM(0, 2) X3_TERM(2, 0) X3_TERM(1, 1) X3_TERM(0, 1) X3_TERM(1, 0) X3_TERM(0, 0)
#undef X3_TERM
                        }
                        if (!(PROBE & 8) && !((CONT == 2 || CONT == 5) && kt + RD >= NK)) bload(ring[ks % RD][j], j, kt + RD);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                { const long long t = tick(); t_mfma += t - t_mark; }
            }
            if constexpr (PF == 2 && !(PROBE & 2)) {
                aload(pre, ps + 1 < passes ? mt : mt + (int)gridDim.x, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            // epilogue
            // (t_mfma: everything since the last chunk's second barrier, all chunks of the pass)
            { const long long t = tick(); t_epi -= t; }
#pragma unroll
            for (int i = 0; i < MR; ++i) {
                const int64_t row = m0 + i * 16 + li;
#pragma unroll
                for (int j = 0; j < WCT; ++j) {
                    int col0 = ncol0 + j * 16 + lg * 4;
                    int64_t rw = row;
                    if (PROBE & 256) {          // tile pair (j, j + 1) as two stores of 8 rows x 128 bytes (the last, unpaired tile as it is)
                        if (j + 1 < WCT || (j & 1)) {
                            rw = m0 + i * 16 + (j & 1) * 8 + (li & 7);
                            col0 = ncol0 + (j & ~1) * 16 + (li >> 3) * 16 + lg * 4;
                        }
                    }
                    if ((PROBE & 1) ? (acc[i][j][0] == 123.4f) : (rw < M && col0 < N)) {
                        if (PROBE & 512) __builtin_nontemporal_store(acc[i][j], reinterpret_cast<f32x4*>(C + rw * ldc + col0));
                        else *reinterpret_cast<f32x4*>(C + rw * ldc + col0) = acc[i][j];
                    }
                }
            }
            { const long long t = tick(); t_epi += t; }
            if (CONT == 5 && !(PROBE & 8)) {          // (debug) the next pass's first fragments requested AFTER this pass's stores
#pragma unroll
                for (int j = 0; j < WCT; ++j) bload(ring[0][j], j, NK);
            }
        }
    }
    if ((PROBE & 128) && tim && blockIdx.x == 7 && lane == 0) {
        const long long t_end = tick();
        tim[wid * 4 + 0] = t_end - t_begin; tim[wid * 4 + 1] = t_load; tim[wid * 4 + 2] = t_epi; tim[wid * 4 + 3] = t_mfma;
    }
}

static int g_loop_reps = 0;          // > 0: every run() launches this many times (clock / power watching: tools/clock_watch.py)
static int g_ncu = 256;              // `loop <ordinal> <n>`: on n CUs only (CU-masked stream, n / 8 per XCD), the grid scaled by n / 256
static hipStream_t g_stream = 0;
static int g_ordinal = 0;
static int g_only = -1;             // >= 0: only the run() with this ordinal

static unsigned short h_bf16_rne(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    const uint32_t lsb = (u >> 16) & 1u;
    u += 0x7fffu + lsb;
    return (unsigned short)(u >> 16);
}
static float h_bf16_f(unsigned short h) {
    const uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

template <int KC, int NCH, int WCT, int PF, int PROBE, int BM = 64, int RD = 1, int STG = 0, int CONT = 0, int AFPF = 0>
static void run(const char* name, const float* dA, int64_t lda, int64_t M, int K, const unsigned short* dB, int N, float* dC,
                int64_t ldc, const std::vector<float>& hA, const std::vector<float>& hW, int grid) {
    const int my = g_ordinal++;
    if (g_only >= 0 && my != g_only) return;
    const int n_mt = (int)((M + BM - 1) / BM);
    grid = grid * g_ncu / 256;
    const int passes = N <= 4 * WCT * 16 ? 1 : 2;
    const size_t lds = (size_t)3 * BM * (KC * 2 + 16);
    static_assert(RD == 1 || (KC / 32) % RD == 0, "ring depth must divide the k-steps of a chunk");
    auto kern = x3_rows_kernel<KC, NCH, WCT, PF, PROBE, BM, RD, STG, CONT, AFPF>;
    static long long* dT = nullptr;
    if (!dT) CK(hipMalloc(&dT, 64 * 8));
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipMemset(dC, 0, (size_t)M * ldc * 4));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, g_stream, dA, lda, M, K, dB, N, dC, ldc, n_mt, passes, dT);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int reps = g_loop_reps ? g_loop_reps : 20;
    CK(hipEventRecord(e0, g_stream));
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, g_stream, dA, lda, M, K, dB, N, dC, ldc, n_mt, passes, dT);
    CK(hipEventRecord(e1, g_stream));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    double worst = 0;
    int nbad = 0;
    const int64_t rows[] = {0, 1, 63, 64, 12345, 512 * 64 + 5, 3 * 512 * 64 + 64 * 7 + 1, M - 65, M - 1};
    std::vector<float> hc(N);
    for (int64_t r : rows) {
        CK(hipMemcpy(hc.data(), dC + r * ldc, (size_t)N * 4, hipMemcpyDeviceToHost));
        for (int n = 0; n < N; ++n) {
            double s = 0, mag = 0;
            for (int k = 0; k < K; ++k) {
                s += (double)hA[r * lda + k] * (double)hW[(size_t)k * N + n];
                mag += fabs((double)hA[r * lda + k] * (double)hW[(size_t)k * N + n]);
            }
            worst = fmax(worst, fabs(s - hc[n]) / (1e-6 + mag));
            if ((CONT == 1 || CONT == 4 || CONT == 5) && !PROBE && fabs(s - hc[n]) / (1e-6 + mag) > 1e-3 && nbad++ < 12) printf("      bad: row %lld col %d got %g want %g\n", (long long)r, n, hc[n], s);
        }
    }
    const double flops = 2.0 * M * N * K;
    printf("[%2d] %-44s grid %4d  passes %d  lds %6zu  %.3f ms  %.1f TF(fp32-equiv)   max err / sum|terms| %.2e\n", my, name, grid, passes, lds, ms,
           flops / ms / 1e9, worst);
    if (PROBE & 128) {
        long long hT[16];
        CK(hipMemcpy(hT, dT, sizeof(hT), hipMemcpyDeviceToHost));
        for (int w = 0; w < 4; ++w)
            printf("    wave %d of block 7: total %lld ticks, load phases %lld (%.0f %%), k loops %lld (%.0f %%), epilogues %lld (%.0f %%)\n", w, hT[w * 4],
                   hT[w * 4 + 1], 100.0 * hT[w * 4 + 1] / hT[w * 4], hT[w * 4 + 3], 100.0 * hT[w * 4 + 3] / hT[w * 4], hT[w * 4 + 2],
                   100.0 * hT[w * 4 + 2] / hT[w * 4]);
    }
}

int main(int argc, char** argv) {
    if (argc >= 3 && !strcmp(argv[1], "loop")) { g_only = atoi(argv[2]); g_loop_reps = 4000; }
    if (argc >= 4 && !strcmp(argv[1], "loop")) {
        g_ncu = atoi(argv[3]);
        g_loop_reps = 4000 * g_ncu / 256;
        uint32_t words[8];
        for (int x = 0; x < 8; ++x) words[x] = g_ncu / 8 >= 32 ? 0xffffffffu : ((1u << (g_ncu / 8)) - 1u);
        CK(hipExtStreamCreateWithCUMask(&g_stream, 8, words));
        printf("CU mask: %d CUs (%d per XCD)\n", g_ncu, g_ncu / 8);
    }
    if (argc >= 3 && !strcmp(argv[1], "pmc")) { g_only = atoi(argv[2]); g_loop_reps = 5; }          // one variant, few launches (rocprofv3 --pmc)
    const int64_t M = 440000;
    const int K = 300;
    const int64_t lda = 300;
    std::vector<float> hA((size_t)M * lda);
    uint32_t s = 12345;
    auto rnd = [&]() {
        s = s * 1664525u + 1013904223u;
        return (float)((s >> 8) & 0xffff) / 65536.f - 0.5f;
    };
    for (auto& x : hA) x = rnd() * (1.0f + 1e-3f * rnd());
    float *dA, *dC;
    CK(hipMalloc(&dA, hA.size() * 4));
    CK(hipMalloc(&dC, (size_t)M * 640 * 4));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    for (int N : {300, 600}) {
        std::vector<float> hW((size_t)K * N);
        for (auto& x : hW) x = rnd() * 0.1f * (1.0f + 1e-3f * rnd());
        const int passes = N <= 320 ? 1 : 2, n_tiles = 4 * passes * 5, NK = 10;
        // fragment order [column tile][k-step][plane][lane][8]
        std::vector<unsigned short> hF((size_t)n_tiles * NK * 3 * 512, 0);
        for (int nt = 0; nt < n_tiles; ++nt)
            for (int kt = 0; kt < NK; ++kt)
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 8; ++e) {
                        const int n = nt * 16 + (lane & 15), k = kt * 32 + (lane >> 4) * 8 + e;
                        if (n < N && k < K) {
                            float x = hW[(size_t)k * N + n];
                            for (int pl = 0; pl < 3; ++pl) {
                                const unsigned short h = h_bf16_rne(x);
                                hF[((((size_t)nt * NK + kt) * 3 + pl) * 64 + lane) * 8 + e] = h;
                                x -= h_bf16_f(h);
                            }
                        }
                    }
        unsigned short* dF;
        CK(hipMalloc(&dF, hF.size() * 2));
        CK(hipMemcpy(dF, hF.data(), hF.size() * 2, hipMemcpyHostToDevice));
        printf("N = %d (%.1f GFLOP)\n", N, 2.0 * M * N * K / 1e9);
        run<160, 2, 5, 0, 0>("x3 rows, 2 chunks of 160", dA, lda, M, K, dF, N, dC, N, hA, hW, 512);
        run<160, 2, 5, 1, 0>("x3 rows, 2 chunks of 160, A prefetched", dA, lda, M, K, dF, N, dC, N, hA, hW, 512);
        run<160, 2, 5, 0, 1>("  ablation: no C stores", dA, lda, M, K, dF, N, dC, N, hA, hW, 512);
        run<160, 2, 5, 0, 2>("  ablation: no A loads", dA, lda, M, K, dF, N, dC, N, hA, hW, 512);
        run<160, 2, 5, 0, 8>("  ablation: no B loads", dA, lda, M, K, dF, N, dC, N, hA, hW, 512);
        run<160, 2, 5, 0, 11>("  ablation: MFMAs only", dA, lda, M, K, dF, N, dC, N, hA, hW, 512);
        run<160, 2, 5, 0, 4>("  ablation: no MFMAs", dA, lda, M, K, dF, N, dC, N, hA, hW, 512);
        run<160, 2, 5, 0, 16>("  probe: one B plane loaded", dA, lda, M, K, dF, N, dC, N, hA, hW, 512);
        run<160, 2, 5, 0, 32>("  probe: B loads hit L1", dA, lda, M, K, dF, N, dC, N, hA, hW, 512);
        run<160, 2, 5, 0, 64>("  probe: no LDS fragment reads", dA, lda, M, K, dF, N, dC, N, hA, hW, 512);
        run<160, 2, 5, 0, 3>("  probe: no A loads, no C stores", dA, lda, M, K, dF, N, dC, N, hA, hW, 512);
        run<160, 2, 5, 0, 0>("  one block per CU", dA, lda, M, K, dF, N, dC, N, hA, hW, 256);
        run<160, 2, 5, 0, 11>("  MFMAs only, one block per CU", dA, lda, M, K, dF, N, dC, N, hA, hW, 256);
        run<160, 2, 5, 0, 3>("  no A loads, no C stores, one block per CU", dA, lda, M, K, dF, N, dC, N, hA, hW, 256);
        run<160, 2, 5, 0, 256>("  probe: whole-line C stores (pattern)", dA, lda, M, K, dF, N, dC, N, hA, hW, 512);
        run<160, 2, 5, 0, 512>("  probe: non-temporal C stores", dA, lda, M, K, dF, N, dC, N, hA, hW, 512);
        run<160, 2, 5, 0, 1024>("  probe: non-temporal A loads", dA, lda, M, K, dF, N, dC, N, hA, hW, 512);
        run<160, 2, 5, 0, 256 + 512 + 1024>("  probe: all three", dA, lda, M, K, dF, N, dC, N, hA, hW, 512);
        run<320, 1, 5, 0, 0>("  one chunk of 320, one block per CU", dA, lda, M, K, dF, N, dC, N, hA, hW, 256);
        run<160, 2, 5, 2, 0, 64, 1, 0, 1>("B ring carried + next pass's first A chunk ahead of the stores", dA, lda, M, K, dF, N, dC, N, hA, hW, 512);
        run<160, 2, 5, 0, 0, 64, 1, 0, 1>("B ring carried across passes and tiles", dA, lda, M, K, dF, N, dC, N, hA, hW, 512);
        run<160, 2, 5, 0, 0, 64, 1, 0, 5>("  carried, requested after the stores (debug)", dA, lda, M, K, dF, N, dC, N, hA, hW, 512);
        run<160, 2, 5, 0, 0, 64, 1, 0, 3>("  carried AND re-filled (debug)", dA, lda, M, K, dF, N, dC, N, hA, hW, 512);
        run<160, 2, 5, 0, 0, 64, 1, 0, 4>("  carried, next tile index through readfirstlane (debug)", dA, lda, M, K, dF, N, dC, N, hA, hW, 512);
        run<160, 2, 5, 0, 0, 64, 1, 0, 2>("  no request past the last k-step, ring re-filled per pass", dA, lda, M, K, dF, N, dC, N, hA, hW, 512);
        run<160, 2, 5, 0, 1, 64, 1, 0, 1>("  ... no C stores", dA, lda, M, K, dF, N, dC, N, hA, hW, 512);
        run<160, 2, 5, 0, 0, 64, 1, 0, 1>("B ring carried across passes and tiles (again)", dA, lda, M, K, dF, N, dC, N, hA, hW, 512);
        run<160, 2, 5, 0, 2048>("  probe: truncation split of A (B planes stay RNE)", dA, lda, M, K, dF, N, dC, N, hA, hW, 512);
        run<160, 2, 5, 0, 0, 64, 1, 1>("  second block of a CU starts 8k cycles late", dA, lda, M, K, dF, N, dC, N, hA, hW, 512);
        run<160, 2, 5, 0, 0, 64, 1, 2>("  second block of a CU starts 16k cycles late", dA, lda, M, K, dF, N, dC, N, hA, hW, 512);
        run<160, 2, 5, 0, 0, 64, 1, 4>("  second block of a CU starts 32k cycles late", dA, lda, M, K, dF, N, dC, N, hA, hW, 512);
        run<160, 2, 5, 0, 0, 64, 1, 8>("  second block of a CU starts 65k cycles late", dA, lda, M, K, dF, N, dC, N, hA, hW, 512);
        run<320, 1, 5, 0, 0, 64, 2>("  one chunk of 320, one block per CU, B two k-steps ahead", dA, lda, M, K, dF, N, dC, N, hA, hW, 256);
        run<320, 1, 5, 0, 3, 64, 2>("    ... no A loads, no C stores", dA, lda, M, K, dF, N, dC, N, hA, hW, 256);
        run<320, 1, 5, 0, 3, 64, 1>("    ... no A loads, no C stores, B one k-step ahead", dA, lda, M, K, dF, N, dC, N, hA, hW, 256);
        run<320, 1, 5, 0, 11, 64, 1>("    ... MFMAs only", dA, lda, M, K, dF, N, dC, N, hA, hW, 256);
        run<160, 2, 5, 0, 0, 128>("128 rows, 2 chunks of 160, one block per CU", dA, lda, M, K, dF, N, dC, N, hA, hW, 256);
        run<160, 2, 5, 0, 11, 128>("  128 rows: MFMAs only", dA, lda, M, K, dF, N, dC, N, hA, hW, 256);
        run<160, 2, 5, 0, 3, 128>("  128 rows: no A loads, no C stores", dA, lda, M, K, dF, N, dC, N, hA, hW, 256);
        run<160, 2, 5, 0, 1, 128>("  128 rows: no C stores", dA, lda, M, K, dF, N, dC, N, hA, hW, 256);
        run<160, 2, 5, 0, 35>("  no A, no C, B loads hit L1", dA, lda, M, K, dF, N, dC, N, hA, hW, 512);
        run<160, 2, 5, 0, 35>("  no A, no C, B loads hit L1, one block per CU", dA, lda, M, K, dF, N, dC, N, hA, hW, 256);
        run<160, 2, 5, 0, 19>("  no A, no C, one B plane", dA, lda, M, K, dF, N, dC, N, hA, hW, 512);
        run<160, 2, 5, 0, 0>("x3 rows, 2 chunks of 160 (again)", dA, lda, M, K, dF, N, dC, N, hA, hW, 512);
        run<160, 2, 5, 0, 0, 128, 1, 0, 0, 1>("128 rows, A fragments one k-step ahead", dA, lda, M, K, dF, N, dC, N, hA, hW, 256);
        run<160, 2, 5, 0, 11, 128, 1, 0, 0, 1>("  ... MFMAs only", dA, lda, M, K, dF, N, dC, N, hA, hW, 256);
        run<160, 2, 5, 0, 3, 128, 1, 0, 0, 1>("  ... no A loads, no C stores", dA, lda, M, K, dF, N, dC, N, hA, hW, 256);
        run<160, 2, 5, 1, 0, 128, 1, 0, 0, 1>("128 rows, A fragments ahead, A chunk prefetched", dA, lda, M, K, dF, N, dC, N, hA, hW, 256);
        run<160, 2, 5, 0, 0, 64, 1, 0, 0, 1>("64 rows, A fragments one k-step ahead", dA, lda, M, K, dF, N, dC, N, hA, hW, 512);
        run<160, 2, 5, 0, 0, 128, 1, 0, 0, 2>("128 rows, half the A fragments a k-step ahead", dA, lda, M, K, dF, N, dC, N, hA, hW, 256);
        run<160, 2, 5, 0, 11, 128, 1, 0, 0, 2>("  ... MFMAs only", dA, lda, M, K, dF, N, dC, N, hA, hW, 256);
        run<160, 2, 5, 0, 3, 128, 1, 0, 0, 2>("  ... no A loads, no C stores", dA, lda, M, K, dF, N, dC, N, hA, hW, 256);
        run<160, 2, 5, 0, 0, 128, 1, 0, 2, 2>("  ... no B request past the last k-step", dA, lda, M, K, dF, N, dC, N, hA, hW, 256);
        CK(hipFree(dF));
    }
    return 0;
}
