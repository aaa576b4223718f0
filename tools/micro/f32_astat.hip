// Micro-benchmark (NOT part of the product): the whole-rows ("A-stationary") kernel of tools/micro/bf16_astat.hip for the EXACT
// fp32 contractions (v_mfma_f32_16x16x4_f32), C[M x N] = A[M x K] . W, M = 440,000, K = 300, N = 300 (one product) or 600 (the
// highway block's dual launch).  64 whole rows of A per block in LDS (fp32: 64 x 308 floats = 79 KB, two blocks per CU), no
// barrier inside a tile, B fragments in fragment order ([column tile][16-k step][lane][4 floats]) straight from L2.  Same MFMA and
// the same k grouping as gemm.hip (lane (li, lg) holds k = 16 s + 4 lg + t for the t-th MFMA of a step): bit-identical results.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/f32_astat.hip -o tools/micro/bin/f32_astat && tools/micro/bin/f32_astat
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e_ = (x);                                                   \
        if (e_ != hipSuccess) {                                                \
            printf("%s: %s\n", #x, hipGetErrorString(e_));                     \
            exit(1);                                                           \
        }                                                                      \
    } while (0)

template <int KP, int BM, int WCT, int DEPTH, int PROBE = 0>
__global__ __launch_bounds__(256, 2) void astat_f32_kernel(const float* __restrict__ A, int64_t lda, int64_t M, int K,
                                                           const float* __restrict__ Bf, int N, float* __restrict__ C, int64_t ldc,
                                                           int n_mt, int passes) {
    constexpr int PITCH = KP + 4;               // floats per LDS row: 77 float4 -> odd -> conflict-free ds_read_b128
    constexpr int F4R = KP / 4;
    constexpr int ITERS = (BM * F4R + 255) / 256;
    constexpr int MR = BM / 16, NK = KP / 16, D1 = DEPTH + 1;
    extern __shared__ __attribute__((aligned(16))) float As[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int K4 = (K + 3) & ~3;
    const uint32_t ld4 = (uint32_t)lda * 4u;
    const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Bf), 0, 4 * passes * WCT * NK * 1024, 0x00020000);
    f32x4 ring[DEPTH + 1][WCT];
    for (int mt = blockIdx.x; mt < n_mt; mt += gridDim.x) {
        const int64_t m0 = (int64_t)mt * BM;
        {
            int tt = tid;
            asm volatile("" : "+v"(tt));
            const int64_t rows_left = M - m0;
            const uint64_t base = reinterpret_cast<uint64_t>(A + m0 * lda);
            const uint64_t bu = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(base >> 32)) << 32) |
                                (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)base);
            const int64_t nbytes = (rows_left < BM ? rows_left : BM) * lda * 4;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(bu), 0,
                                                                               __builtin_amdgcn_readfirstlane((int)nbytes), 0x00020000);
            f32x4 v[ITERS];
#pragma unroll
            for (int i = 0; i < ((PROBE & 2) ? 0 : ITERS); ++i) {
                const int idx = tt + 256 * i;
                const int r = idx / F4R, c = idx - r * F4R;
                const uint32_t off = (idx < BM * F4R && c * 4 < K4) ? (uint32_t)r * ld4 + (uint32_t)c * 16u : 0x80000000u;
                v[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0));
            }
#pragma unroll
            for (int i = 0; i < ((PROBE & 2) ? 0 : ITERS); ++i) {
                const int idx = tt + 256 * i;
                const int r = idx / F4R, c = idx - r * F4R;
                if (idx < BM * F4R) *reinterpret_cast<f32x4*>(As + r * PITCH + c * 4) = v[i];
            }
        }
        __syncthreads();
#pragma unroll 1
        for (int ps = 0; ps < passes; ++ps) {
            const int tile0 = (wid * passes + ps) * WCT;
            const int ncol0 = tile0 * 16;
            f32x4 acc[MR][WCT];
#pragma unroll
            for (int i = 0; i < MR; ++i)
#pragma unroll
                for (int j = 0; j < WCT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            auto bload_at = [&](f32x4 (&b)[WCT], int t0, int kt) {
                const int kk = kt < NK ? kt : NK - 1;
#pragma unroll
                for (int j = 0; j < WCT; ++j)
                    b[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(brs, lane * 16, ((t0 + j) * NK + kk) * 1024, 0));
            };
            auto bload = [&](f32x4 (&b)[WCT], int kt) { bload_at(b, tile0, kt); };
            auto kstep = [&](const f32x4 (&b)[WCT], int kt) {
                f32x4 af[MR];
#pragma unroll
                for (int i = 0; i < MR; ++i) af[i] = *reinterpret_cast<const f32x4*>(As + (i * 16 + li) * PITCH + kt * 16 + lg * 4);
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int j = 0; j < WCT; ++j)
#pragma unroll
                        for (int i = 0; i < MR; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[j][t], af[i][t], acc[i][j], 0, 0, 0);
            };
            // PROBE 16 ("early"): the first DEPTH fragment sets of a pass are requested BEFORE the stores of the pass in front of it
            // (gfx9 retires loads and stores through ONE in-order counter: a fragment requested behind 20 stores is not usable
            // until every one of them has been acknowledged; requested ahead of them it is, and DEPTH steps of MFMAs -- 2 x 2,560
            // cycles -- then run while the stores drain)
            if (!(PROBE & 16) || (ps == 0 && mt == (int)blockIdx.x)) {
#pragma unroll
                for (int d = 0; d < DEPTH; ++d) bload(ring[d], d);
            }
#pragma unroll 1
            for (int k0 = 0; k0 < NK; k0 += D1) {
#pragma unroll
                for (int u = 0; u < D1; ++u) {
                    bload(ring[(u + DEPTH) % D1], k0 + u + DEPTH);
                    if (k0 + u < NK) kstep(ring[u], k0 + u);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (PROBE & 16) {
                const int nt0 = (wid * passes + (ps + 1 < passes ? ps + 1 : 0)) * WCT;
#pragma unroll
                for (int d = 0; d < DEPTH; ++d) bload_at(ring[d], nt0, d);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < MR; ++i) {
                const int64_t row = m0 + i * 16 + li;
#pragma unroll
                for (int j = 0; j < WCT; ++j) {
                    const int col0 = ncol0 + j * 16 + lg * 4;
                    if ((PROBE & 4) ? (acc[i][j][0] == 123.4f) : (row < M && col0 < N)) {
                        if (PROBE & 8) __builtin_nontemporal_store(acc[i][j], reinterpret_cast<f32x4*>(C + row * ldc + col0));   // PROBE 8
                        else *reinterpret_cast<f32x4*>(C + row * ldc + col0) = acc[i][j];
                    }
                }
            }
        }
        __syncthreads();
    }
}

// Variant: the NEXT tile's rows are requested before this tile's MFMA loop and sit in registers (19 float4 per thread) until the
// tile is done; then one barrier, 19 LDS writes, one barrier.  The HBM read of A leaves the critical path.
template <int KP, int BM, int WCT, int DEPTH>
__global__ __launch_bounds__(256, 2) void astat_f32_prefetch_kernel(const float* __restrict__ A, int64_t lda, int64_t M, int K,
                                                                    const float* __restrict__ Bf, int N, float* __restrict__ C,
                                                                    int64_t ldc, int n_mt, int passes) {
    constexpr int PITCH = KP + 4;
    constexpr int F4R = KP / 4;
    constexpr int ITERS = BM * F4R / 256;
    constexpr int MR = BM / 16, NK = KP / 16, D1 = DEPTH + 1;
    extern __shared__ __attribute__((aligned(16))) float As[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int K4 = (K + 3) & ~3;
    const uint32_t ld4 = (uint32_t)lda * 4u;
    const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Bf), 0, 4 * passes * WCT * NK * 1024, 0x00020000);
    auto tile_rsrc = [&](int mt) {
        const int64_t m0 = (int64_t)mt * BM;
        const int64_t rows_left = mt < n_mt ? M - m0 : 0;
        const uint64_t base = reinterpret_cast<uint64_t>(A + (mt < n_mt ? m0 : 0) * lda);
        const uint64_t bu = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(base >> 32)) << 32) |
                            (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)base);
        const int64_t nbytes = (rows_left < BM ? rows_left : BM) * lda * 4;
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(bu), 0, __builtin_amdgcn_readfirstlane((int)nbytes), 0x00020000);
    };
    auto load_tile = [&](f32x4 (&v)[ITERS], int mt) {
        int tt = tid;
        asm volatile("" : "+v"(tt));
        const __amdgpu_buffer_rsrc_t rs = tile_rsrc(mt);
#pragma unroll
        for (int i = 0; i < ITERS; ++i) {
            const int idx = tt + 256 * i;
            const int r = idx / F4R, c = idx - r * F4R;
            v[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(c * 4 < K4 ? (uint32_t)r * ld4 + (uint32_t)c * 16u : 0x80000000u), 0, 0));
        }
    };
    auto store_tile = [&](const f32x4 (&v)[ITERS]) {
        int tt = tid;
        asm volatile("" : "+v"(tt));
#pragma unroll
        for (int i = 0; i < ITERS; ++i) {
            const int idx = tt + 256 * i;
            const int r = idx / F4R, c = idx - r * F4R;
            *reinterpret_cast<f32x4*>(As + r * PITCH + c * 4) = v[i];
        }
    };
    f32x4 v[ITERS];
    load_tile(v, blockIdx.x);
    store_tile(v);
    __syncthreads();
    for (int mt = blockIdx.x; mt < n_mt; mt += gridDim.x) {
        const int64_t m0 = (int64_t)mt * BM;
        load_tile(v, mt + gridDim.x);                 // (zeros past the end of the list)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
        for (int ps = 0; ps < passes; ++ps) {
            const int tile0 = (wid * passes + ps) * WCT;
            const int ncol0 = tile0 * 16;
            f32x4 acc[MR][WCT];
#pragma unroll
            for (int i = 0; i < MR; ++i)
#pragma unroll
                for (int j = 0; j < WCT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
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
                    for (int j = 0; j < WCT; ++j)
#pragma unroll
                        for (int i = 0; i < MR; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[j][t], af[i][t], acc[i][j], 0, 0, 0);
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
#pragma unroll
            for (int i = 0; i < MR; ++i) {
                const int64_t row = m0 + i * 16 + li;
#pragma unroll
                for (int j = 0; j < WCT; ++j) {
                    const int col0 = ncol0 + j * 16 + lg * 4;
                    if (row < M && col0 < N) *reinterpret_cast<f32x4*>(C + row * ldc + col0) = acc[i][j];
                }
            }
        }
        __syncthreads();
        store_tile(v);
        __syncthreads();
    }
}

template <int KP, int BM, int WCT, int DEPTH, int PROBE>
static void run(const char* name, const float* dA, int64_t lda, int64_t M, int K, const float* dB, int N, float* dC, int64_t ldc,
                const std::vector<float>& hA, const std::vector<float>& hW, int grid) {
    const int n_mt = (int)((M + BM - 1) / BM);
    const int passes = N <= 4 * WCT * 16 ? 1 : 2;
    const size_t lds = (size_t)BM * (KP + 4) * 4;
    auto kern = astat_f32_kernel<KP, BM, WCT, DEPTH, PROBE>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipMemset(dC, 0, (size_t)M * ldc * 4));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, dA, lda, M, K, dB, N, dC, ldc, n_mt, passes);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int reps = 20;
    CK(hipEventRecord(e0));
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, dA, lda, M, K, dB, N, dC, ldc, n_mt, passes);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    double worst = 0;
    const int64_t rows[] = {0, 1, 63, 64, 12345, M - 65, M - 1};
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
        }
    }
    const double flops = 2.0 * M * N * K;
    printf("%-40s grid %4d  passes %d  %.3f ms  %.1f TF   max err / sum|terms| %.2e\n", name, grid, passes, ms, flops / ms / 1e9, worst);
}

// ---- round 4: NO LDS at all -- every wave streams its own A fragments ------------------------------------------------------------
// The A fragment of the swapped-operand MFMA -- lane (li, lg) holds A[row = li][k = 4 lg .. 4 lg + 3] of a 16 x 16 block -- is one
// buffer_load_dwordx4 from row-major A (16 rows x 64 contiguous bytes per wave instruction).  So a wave can take its MR row
// fragments and WCT weight fragments per 16 k straight into registers, DEPTH steps ahead, ACROSS tile boundaries (the k loop is
// flattened over the wave's tiles): no LDS image, no barrier, no load / multiply / store phases -- 8 independent streams per CU,
// whose loads, MFMAs and C stores overlap because they belong to different waves.  The price: the waves that share a row tile
// (one per 80-column group) each fetch its A fragments -- the second to eighth time from the CU's L1.
template <int KP, int WCT, int DEPTH, int PROBE = 0, int NW = 8>
__global__ __launch_bounds__(64 * NW, 1) void direct_f32_kernel(const float* __restrict__ A, int64_t lda, int64_t M, int K,
                                                            const float* __restrict__ Bf, int N, float* __restrict__ C, int64_t ldc,
                                                            int n_mt, int passes) {
    constexpr int MR = 4, NK = KP / 16, D = DEPTH;
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wid >= 8) return;                             // (NW > 8: the extra waves only cap the register budget in this probe)
    const int li = lane & 15, lg = lane >> 4;
    const int n_groups = 4 * passes;                  // 80-column groups: 4 (N <= 320) or 8
    const int cg = wid % n_groups, sub = wid / n_groups, n_sub = 8 / n_groups;       // row tiles a block works on at a time: 2 or 1
    const int tile0 = cg * WCT, ncol0 = tile0 * 16;
    const int K4 = (K + 3) & ~3;
    // this wave's row tiles: (blockIdx * n_sub + sub) + j * gridDim * n_sub
    const int first = (int)blockIdx.x * n_sub + sub, stride = (int)gridDim.x * n_sub;
    const int n_my = first < n_mt ? (n_mt - 1 - first) / stride + 1 : 0;
    if (n_my == 0) return;
    const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A), 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Bf), 0, 4 * passes * WCT * NK * 1024, 0x00020000);
    const uint32_t ld4 = (uint32_t)lda * 4u;
    f32x4 ra[D + 1][MR], rb[D + 1][WCT];
    const int total = n_my * NK;
    auto fetch = [&](f32x4 (&fa)[MR], f32x4 (&fb)[WCT], int g) {          // global step g = (tile index) * NK + kt
        const int gg = g < total ? g : total - 1;
        const int ti = gg / NK, kt = gg - ti * NK;
        const int64_t m0 = (int64_t)(first + ti * stride) * 64;
        const bool k_ok = kt * 16 + lg * 4 < K4;
#pragma unroll
        for (int i = 0; i < MR; ++i) {
            const int64_t row = m0 + i * 16 + li;
            const uint32_t off = (row < M && k_ok) ? (uint32_t)row * ld4 + (uint32_t)(kt * 64 + lg * 16) : 0x80000000u;
            fa[i] = (PROBE & 2) ? f32x4{1.f, 1.f, 1.f, 1.f} : __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ars, (int)off, 0, 0));
        }
#pragma unroll
        for (int j = 0; j < WCT; ++j)
            fb[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(brs, lane * 16, ((tile0 + j) * NK + kt) * 1024, 0));
    };
    f32x4 acc[MR][WCT];
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < WCT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (PROBE & 64) {
        // PROBE 64: the 8 waves of a CU start an eighth of a tile apart (a tile is 19 k-steps of ~5,100 cycles), so that their
        // epilogues -- 20 row-scattered stores each, which the CU's store path takes at ~7 B/clk -- never coincide
        for (int i = 0; i < wid * ((PROBE >> 8) & 0xff); ++i) __builtin_amdgcn_s_sleep(127);
    }
#pragma unroll
    for (int d = 0; d < D; ++d) fetch(ra[d], rb[d], d);
    int kt = 0, ti = 0;
    uint64_t t_epi0 = 0, t_after = 0, t_mid = 0, c_epi = 0, c_next = 0, c_mid = 0;
    int n_epi = 0;
#pragma unroll 1
    for (int g0 = 0; g0 < total; g0 += D + 1) {
#pragma unroll
        for (int u = 0; u <= D; ++u) {
            fetch(ra[(u + D) % (D + 1)], rb[(u + D) % (D + 1)], g0 + u + D);
            if (g0 + u < total) {
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int j = 0; j < WCT; ++j)
#pragma unroll
                        for (int i = 0; i < MR; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(rb[u][j][t], ra[u][i][t], acc[i][j], 0, 0, 0);
                if (++kt == NK) {              // tile done: store, clear (the next tile's fragments are already on their way)
                    const int64_t m0 = (int64_t)(first + ti * stride) * 64;
                    if (PROBE & 32) { t_epi0 = __builtin_readcyclecounter(); }
#pragma unroll
                    for (int i = 0; i < MR; ++i) {
                        const int64_t row = m0 + i * 16 + li;
#pragma unroll
                        for (int j = 0; j < WCT; ++j) {
                            const int col0 = ncol0 + j * 16 + lg * 4;
                            if (PROBE & 16) {           // PROBE 16: the tile goes to this wave's 20 KB of LDS instead of to memory
                                extern __shared__ __attribute__((aligned(16))) float Cs[];
                                *reinterpret_cast<f32x4*>(Cs + wid * 5120 + ((i * 16 + li) * 80 + j * 16 + lg * 4)) = acc[i][j];
                                acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                                continue;
                            }
                            // PROBE 8: every tile is written to the first rows of C (the stores stay in L2: no HBM write traffic)
                            const int64_t wrow = (PROBE & 8) ? (i * 16 + li + 64 * (blockIdx.x & 63)) : row;
                            if ((PROBE & 4) ? (acc[i][j][0] == 123.4f) : (row < M && col0 < N)) *reinterpret_cast<f32x4*>(C + wrow * ldc + col0) = acc[i][j];
                            acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                        }
                    }
                    kt = 0;
                    ++ti;
                    if (PROBE & 32) { const uint64_t t1 = __builtin_readcyclecounter(); c_epi += t1 - t_epi0; t_after = t1; n_epi++; }
                } else if ((PROBE & 32) && kt == 1 && t_after) {
                    c_next += __builtin_readcyclecounter() - t_after;        // the first k-step after an epilogue
                    t_after = 0;
                } else if ((PROBE & 32) && kt == 9) {
                    t_mid = __builtin_readcyclecounter();
                } else if ((PROBE & 32) && kt == 10) {
                    c_mid += __builtin_readcyclecounter() - t_mid;           // a k-step in the middle of a tile
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if ((PROBE & 32) && lane == 0 && blockIdx.x == 7) {
        float* o = C + (int64_t)wid * 8;
        o[0] = (float)c_epi / n_epi; o[1] = (float)c_next / n_epi; o[2] = (float)c_mid / n_epi; o[3] = (float)n_epi;
    }
}

template <int KP, int WCT, int DEPTH, int PROBE, int NW = 8>
static void run_direct(const char* name, const float* dA, int64_t lda, int64_t M, int K, const float* dB, int N, float* dC, int64_t ldc,
                       const std::vector<float>& hA, const std::vector<float>& hW, int grid) {
    const int n_mt = (int)((M + 63) / 64);
    const int passes = N <= 4 * WCT * 16 ? 1 : 2;
    auto kern = direct_f32_kernel<KP, WCT, DEPTH, PROBE, NW>;
    CK(hipMemset(dC, 0, (size_t)M * ldc * 4));
    const size_t lds = (PROBE & 16) ? 8 * 5120 * 4 : 0;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), lds, 0, dA, lda, M, K, dB, N, dC, ldc, n_mt, passes);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int reps = 20;
    CK(hipEventRecord(e0));
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), lds, 0, dA, lda, M, K, dB, N, dC, ldc, n_mt, passes);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    double worst = 0;
    const int64_t rows[] = {0, 1, 63, 64, 127, 128, 12345, 64 * 256 * 5 + 17, 64 * 511 + 3, M - 65, M - 1};
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
        }
    }
    const double flops = 2.0 * M * N * K;
    printf("%-40s grid %4d  passes %d  %.3f ms  %.1f TF   max err / sum|terms| %.2e\n", name, grid, passes, ms, flops / ms / 1e9, worst);
    if (PROBE & 32) {
        float h[64];
        CK(hipMemcpy(h, dC, sizeof(h), hipMemcpyDeviceToHost));
        for (int w = 0; w < 8; ++w)
            printf("    wave %d of block 7: epilogue %.0f cycles, first k-step after it %.0f, a mid-tile k-step %.0f  (%d tiles; s_memtime ticks)\n", w,
                   h[w * 8], h[w * 8 + 1], h[w * 8 + 2], (int)h[w * 8 + 3]);
    }
}

template <int KP, int BM, int WCT, int DEPTH>
static void run_pf(const char* name, const float* dA, int64_t lda, int64_t M, int K, const float* dB, int N, float* dC, int64_t ldc, int grid) {
    const int n_mt = (int)((M + BM - 1) / BM);
    const int passes = N <= 4 * WCT * 16 ? 1 : 2;
    const size_t lds = (size_t)BM * (KP + 4) * 4;
    auto kern = astat_f32_prefetch_kernel<KP, BM, WCT, DEPTH>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, dA, lda, M, K, dB, N, dC, ldc, n_mt, passes);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int reps = 20;
    CK(hipEventRecord(e0));
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, dA, lda, M, K, dB, N, dC, ldc, n_mt, passes);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    printf("%-40s grid %4d  passes %d  %.3f ms  %.1f TF\n", name, grid, passes, ms, 2.0 * M * N * K / ms / 1e9);
}

int main() {
    const int64_t M = 440000;
    const int K = 300, KP = 304;
    const int64_t lda = 300;
    std::vector<float> hA((size_t)M * lda);
    uint32_t s = 12345;
    auto rnd = [&]() {
        s = s * 1664525u + 1013904223u;
        return (float)((s >> 8) & 0xffff) / 65536.f - 0.5f;
    };
    for (auto& x : hA) x = rnd();
    float *dA, *dC;
    CK(hipMalloc(&dA, hA.size() * 4));
    CK(hipMalloc(&dC, (size_t)M * 640 * 4));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    for (int N : {300, 600}) {
        std::vector<float> hW((size_t)K * N);
        for (auto& x : hW) x = rnd() * 0.1f;
        const int passes = N <= 320 ? 1 : 2, n_tiles = 4 * passes * 5, NK = KP / 16;
        std::vector<float> hF((size_t)n_tiles * NK * 256, 0.f);
        for (int nt = 0; nt < n_tiles; ++nt)
            for (int kt = 0; kt < NK; ++kt)
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 4; ++e) {
                        const int n = nt * 16 + (lane & 15), k = kt * 16 + (lane >> 4) * 4 + e;
                        if (n < N && k < K) hF[(((size_t)nt * NK + kt) * 64 + lane) * 4 + e] = hW[(size_t)k * N + n];
                    }
        float* dF;
        CK(hipMalloc(&dF, hF.size() * 4));
        CK(hipMemcpy(dF, hF.data(), hF.size() * 4, hipMemcpyHostToDevice));
        printf("N = %d (%.1f GFLOP)\n", N, 2.0 * M * N * K / 1e9);
        run<304, 64, 5, 2, 0>("whole rows, B 2 steps ahead", dA, lda, M, K, dF, N, dC, N, hA, hW, 512);
        run_direct<304, 5, 3, 0>("no LDS: direct fragments, 3 ahead", dA, lda, M, K, dF, N, dC, N, hA, hW, 256);
        run_direct<304, 5, 3, 16>("  C tile into LDS, 3 ahead (8 waves)", dA, lda, M, K, dF, N, dC, N, hA, hW, 256);
        run_direct<304, 5, 1, 16>("  C tile into LDS, 1 ahead (8 waves)", dA, lda, M, K, dF, N, dC, N, hA, hW, 256);
        run_direct<304, 5, 1, 16, 10>("  C tile into LDS, 1 ahead, 10-wave budget", dA, lda, M, K, dF, N, dC, N, hA, hW, 256);
        run_direct<304, 5, 1, 4, 10>("  no C stores, 1 ahead, 10-wave budget", dA, lda, M, K, dF, N, dC, N, hA, hW, 256);
        run<304, 64, 5, 2, 0>("whole rows, B 2 steps ahead (again)", dA, lda, M, K, dF, N, dC, N, hA, hW, 512);
        CK(hipFree(dF));
    }
    return 0;
}
