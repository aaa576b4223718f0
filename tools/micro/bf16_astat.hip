// Micro-benchmark (NOT part of the product): an "A-stationary" bf16 GEMM for the skinny contractions of the bf16 configuration
// (C[M x N] = bf16(A)[M x K] . bf16(W), M = 440,000, N = K = 600, fp32 accumulate).  The library kernel (gemm_bf16.hip) stages
// 128 x 32 slices of A per barrier: one 128-byte line per row and stage, 26 KB in flight per block -- latency-bound at 15 % of
// the bf16 pipe.  Here a block takes BM = 64 WHOLE rows of A (one contiguous 154 KB read, all loads issued at once), converts
// them to bf16 into LDS once (79 KB: two blocks per CU), and every wave multiplies all 64 rows by its own 160 columns with B
// fragments read straight from the L2-resident weight planes [N][Kp] -- no per-stage barrier, A read from HBM exactly once.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/bf16_astat.hip -o gpurun_out/bf16_astat && gpurun_out/bf16_astat
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t bf16_pack(float lo, float hi) {
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e_ = (x);                                                   \
        if (e_ != hipSuccess) {                                                \
            printf("%s: %s\n", #x, hipGetErrorString(e_));                     \
            exit(1);                                                           \
        }                                                                      \
    } while (0)

// WCT column tiles per wave and PASS (acc = MR x WCT x 4 registers), PASSES passes per wave over its columns (the A fragments are
// re-read from LDS in every pass), B fragments requested DEPTH k-steps ahead (register ring, k loop fully unrolled)
template <int KP, int BM, int WCT, int PASSES, int DEPTH, int FR, int PROBE = 0>
__global__ __launch_bounds__(256, BM <= 32 ? 4 : 2) void astat_kernel(const float* __restrict__ A, int64_t lda, int64_t M, int K,
                                                       const unsigned short* __restrict__ Bp, int N, float* __restrict__ C,
                                                       int64_t ldc, int n_mt) {
    constexpr int PITCH = KP * 2 + 16;          // bytes per LDS row: odd multiple of 16 -> conflict-free ds_read_b128
    constexpr int F4R = KP / 4;                 // float4 per row
    constexpr int ITERS = BM * F4R / 256;
    constexpr int MR = BM / 16;
    constexpr int NK = KP / 32;
    static_assert(BM * F4R % 256 == 0, "tile does not divide over 256 threads");
    extern __shared__ __attribute__((aligned(16))) unsigned char As[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int K4 = (K + 3) & ~3;
    const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(Bp), 0, 640 * KP * 2, 0x00020000);
    if ((PROBE & 8) && blockIdx.x >= gridDim.x / 2) {
        // PROBE 8: the second block of every CU starts half a tile late, so that its load / store phases fall into the
        // other block's MFMA phase (blocks that start together stay in lockstep)
        for (int i = 0; i < (PROBE >> 4); ++i) __builtin_amdgcn_s_sleep(127);
    }
    for (int mt = blockIdx.x; mt < n_mt; mt += gridDim.x) {
        const int64_t m0 = (int64_t)mt * BM;
        {
            // (the per-thread offsets of the 38 loads and stores are the same for every tile: left alone, hipcc computes them once
            //  and keeps ~76 registers alive across the MFMA loop -- an opaque copy of the thread index makes them per-tile work)
            int tt = tid;
            asm volatile("" : "+v"(tt));
            // one descriptor per tile: rows past M read as zeros in hardware, offsets are 32-bit
            const int64_t rows_left = M - m0;
            const uint64_t base = reinterpret_cast<uint64_t>(A + m0 * lda);
            const uint64_t bu = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(base >> 32)) << 32) |
                                (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)base);
            const int64_t nbytes = (rows_left < BM ? rows_left : BM) * lda * 4;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(bu), 0,
                                                                               __builtin_amdgcn_readfirstlane((int)nbytes), 0x00020000);
            const uint32_t ld4 = (uint32_t)lda * 4u;
            constexpr int CH = ITERS % 2 == 0 ? 2 : 1, IPC = ITERS / CH;        // two chunks of loads: half the staging registers
#pragma unroll
            for (int ch = 0; ch < ((PROBE & 2) ? 0 : CH); ++ch) {       // PROBE 2: no A loads (LDS holds garbage)
                float4 v[IPC];
#pragma unroll
                for (int i = 0; i < IPC; ++i) {
                    const int idx = tt + 256 * (ch * IPC + i);
                    const int r = idx / F4R, c = idx - r * F4R;
                    const uint32_t off = c * 4 < K4 ? (uint32_t)r * ld4 + (uint32_t)c * 16u : 0x80000000u;
                    const f32x4 t = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0));
                    v[i] = make_float4(t.x, t.y, t.z, t.w);
                }
#pragma unroll
                for (int i = 0; i < IPC; ++i) {
                    const int idx = tt + 256 * (ch * IPC + i);
                    const int r = idx / F4R, c = idx - r * F4R;
                    uint2 w;
                    w.x = bf16_pack(v[i].x, v[i].y);
                    w.y = bf16_pack(v[i].z, v[i].w);
                    *reinterpret_cast<uint2*>(As + r * PITCH + c * 8) = w;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
#pragma unroll 1
        for (int ps = 0; ps < PASSES; ++ps) {
            const int ncol0 = (wid * PASSES + ps) * WCT * 16;
            const uint32_t bvoff = (uint32_t)(li * KP + lg * 8) * 2u;       // lane part; the rest is scalar
            f32x4 acc[MR][WCT];
#pragma unroll
            for (int i = 0; i < MR; ++i)
#pragma unroll
                for (int j = 0; j < WCT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            // B fragments DEPTH k-steps ahead in a ring of DEPTH + 1 register sets.  The loop stays ROLLED (DEPTH + 1 k-steps per
            // trip, ring slots are compile-time constants inside a trip) so that the scheduler cannot hoist the whole tile's
            // loads in front of the first MFMA; loads past the last k-step re-read the last one (no branch around a load)
            constexpr int D1 = DEPTH + 1;
            auto bload = [&](bf16x8 (&b)[WCT], int kt) {
                const int kk = (PROBE & 1) ? 0 : (kt < NK ? kt : NK - 1);      // PROBE 1: one (L1-resident) fragment set
#pragma unroll
                for (int j = 0; j < WCT; ++j)
                    b[j] = FR ? __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
                                    brs, lane * 16, ((ncol0 / 16 + j) * NK + kk) * 1024, 0))
                              : __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
                                    brs, (int)bvoff, ((ncol0 + j * 16) * KP + kk * 32) * 2, 0));
            };
            auto kstep = [&](const bf16x8 (&b)[WCT], int kt) {
                bf16x8 af[MR];
#pragma unroll
                for (int i = 0; i < MR; ++i) af[i] = *reinterpret_cast<const bf16x8*>(As + (i * 16 + li) * PITCH + kt * 64 + lg * 16);
#pragma unroll
                for (int j = 0; j < WCT; ++j)
#pragma unroll
                    for (int i = 0; i < MR; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], af[i], acc[i][j], 0, 0, 0);
            };
            bf16x8 ring[D1][WCT];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) bload(ring[d], d);
#pragma unroll 1
            for (int k0 = 0; k0 < NK; k0 += D1) {
#pragma unroll
                for (int u = 0; u < D1; ++u) {
                    bload(ring[(u + DEPTH) % D1], k0 + u + DEPTH);
                    if (k0 + u < NK) kstep(ring[u], k0 + u);
                    __builtin_amdgcn_sched_barrier(0);       // keep the requests where they are written: one set per k-step
                }
            }
            // swapped operands: lane (li, lg) owns C[row = li][4 lg .. 4 lg + 3] of each 16 x 16 tile
#pragma unroll
            for (int i = 0; i < MR; ++i) {
                const int64_t row = m0 + i * 16 + li;
#pragma unroll
                for (int j = 0; j < WCT; ++j) {
                    const int col0 = ncol0 + j * 16 + lg * 4;
                    if ((PROBE & 4) ? (acc[i][j][0] == 123.4f) : (row < M && col0 < N))      // PROBE 4: no C stores
                        *reinterpret_cast<float4*>(C + row * ldc + col0) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
                }
            }
        }
        __syncthreads();
    }
}

// Software-pipelined form: the NEXT tile's rows are requested while this tile is multiplied (half before each column pass),
// packed to bf16 in registers as they arrive, and written to LDS between two barriers when the tile is done -- the HBM read of A
// no longer sits between the MFMA phases.  Two column passes of WCT tiles per wave (N <= 4 waves x 2 x WCT x 16).
template <int KP, int BM, int WCT, int DEPTH, int PROBE = 0>
__global__ __launch_bounds__(256, 2) void astat_pipe_kernel(const float* __restrict__ A, int64_t lda, int64_t M, int K,
                                                            const unsigned short* __restrict__ Bp, int N, float* __restrict__ C,
                                                            int64_t ldc, int n_mt) {
    constexpr int PITCH = KP * 2 + 16;
    constexpr int F4R = KP / 4;
    constexpr int ITERS = BM * F4R / 256;
    constexpr int IPC = ITERS / 2;
    constexpr int MR = BM / 16;
    constexpr int NK = KP / 32;
    constexpr int D1 = DEPTH + 1;
    static_assert(BM * F4R % 512 == 0, "tile does not divide over 256 threads in two chunks");
    extern __shared__ __attribute__((aligned(16))) unsigned char As[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int li = lane & 15;
    const int lg = lane >> 4;
    const int K4 = (K + 3) & ~3;
    const uint32_t ld4 = (uint32_t)lda * 4u;
    const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(Bp), 0, 640 * KP * 2, 0x00020000);
    auto tile_rsrc = [&](int mt) {
        // rows past M -- and whole tiles past the end of this block's list -- read as zeros in hardware
        const int64_t m0 = (int64_t)mt * BM;
        const int64_t rows_left = mt < n_mt ? M - m0 : 0;
        const uint64_t base = reinterpret_cast<uint64_t>(A + (mt < n_mt ? m0 : 0) * lda);
        const uint64_t bu = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(base >> 32)) << 32) |
                            (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)base);
        const int64_t nbytes = (rows_left < BM ? rows_left : BM) * lda * 4;
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(bu), 0, __builtin_amdgcn_readfirstlane((int)nbytes), 0x00020000);
    };
    auto load_chunk = [&](float4 (&v)[IPC], __amdgpu_buffer_rsrc_t rs, int ch) {
        int tt = tid;
        asm volatile("" : "+v"(tt));            // (keeps the 19 offsets per-use work instead of 19 live registers)
#pragma unroll
        for (int i = 0; i < IPC; ++i) {
            const int idx = tt + 256 * (ch * IPC + i);
            const int r = idx / F4R, c = idx - r * F4R;
            const uint32_t off = c * 4 < K4 ? (uint32_t)r * ld4 + (uint32_t)c * 16u : 0x80000000u;
            const f32x4 t = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0));
            v[i] = make_float4(t.x, t.y, t.z, t.w);
        }
    };
    auto pack_chunk = [&](const float4 (&v)[IPC], uint2 (&pk)[ITERS], int ch) {
#pragma unroll
        for (int i = 0; i < IPC; ++i) {
            pk[ch * IPC + i].x = bf16_pack(v[i].x, v[i].y);
            pk[ch * IPC + i].y = bf16_pack(v[i].z, v[i].w);
        }
    };
    auto store_tile = [&](const uint2 (&pk)[ITERS]) {
        int tt = tid;
        asm volatile("" : "+v"(tt));
#pragma unroll
        for (int i = 0; i < ITERS; ++i) {
            const int idx = tt + 256 * i;
            const int r = idx / F4R, c = idx - r * F4R;
            *reinterpret_cast<uint2*>(As + r * PITCH + c * 8) = pk[i];
        }
    };
    uint2 pk[ITERS];
    {
        const __amdgpu_buffer_rsrc_t rs = tile_rsrc(blockIdx.x);
        float4 v0[IPC], v1[IPC];
        load_chunk(v0, rs, 0);
        load_chunk(v1, rs, 1);
        pack_chunk(v0, pk, 0);
        pack_chunk(v1, pk, 1);
        store_tile(pk);
    }
    __syncthreads();
    for (int mt = blockIdx.x; mt < n_mt; mt += gridDim.x) {
        const int64_t m0 = (int64_t)mt * BM;
        const __amdgpu_buffer_rsrc_t nrs = tile_rsrc(mt + gridDim.x);
        float4 v[IPC];
        if (!(PROBE & 2)) load_chunk(v, nrs, 0);
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            const int ncol0 = (wid * 2 + ps) * WCT * 16;
            f32x4 acc[MR][WCT];
#pragma unroll
            for (int i = 0; i < MR; ++i)
#pragma unroll
                for (int j = 0; j < WCT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            auto bload = [&](bf16x8 (&b)[WCT], int kt) {
                const int kk = kt < NK ? kt : NK - 1;
#pragma unroll
                for (int j = 0; j < WCT; ++j)
                    b[j] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(brs, lane * 16, ((ncol0 / 16 + j) * NK + kk) * 1024, 0));
            };
            auto kstep = [&](const bf16x8 (&b)[WCT], int kt) {
                bf16x8 af[MR];
#pragma unroll
                for (int i = 0; i < MR; ++i) af[i] = *reinterpret_cast<const bf16x8*>(As + (i * 16 + li) * PITCH + kt * 64 + lg * 16);
#pragma unroll
                for (int j = 0; j < WCT; ++j)
#pragma unroll
                    for (int i = 0; i < MR; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], af[i], acc[i][j], 0, 0, 0);
            };
            bf16x8 ring[D1][WCT];
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
            // the half of the next tile requested before this pass has arrived long ago: pack it, request the other half
            if (!(PROBE & 2)) {
                pack_chunk(v, pk, ps);
                if (ps == 0) load_chunk(v, nrs, 1);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < MR; ++i) {
                const int64_t row = m0 + i * 16 + li;
#pragma unroll
                for (int j = 0; j < WCT; ++j) {
                    const int col0 = ncol0 + j * 16 + lg * 4;
                    if ((PROBE & 4) ? (acc[i][j][0] == 123.4f) : (row < M && col0 < N))
                        *reinterpret_cast<float4*>(C + row * ldc + col0) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        store_tile(pk);
        __syncthreads();
    }
}

template <int KP, int BM, int WCT, int DEPTH, int PROBE = 0>
static void run_pipe(const char* name, const float* dA, int64_t lda, int64_t M, int K, const unsigned short* dB, int N, float* dC,
                     int64_t ldc, const std::vector<float>& hA, const std::vector<unsigned short>& hB, int grid);

// Variant: the first half of the NEXT tile's rows is requested after the last k-step of this tile, BEFORE its epilogue's stores
// (gfx9 counts loads and stores in one vmcnt: a load issued behind 80 stores waits for all of them), and the next column pass's
// first B fragments likewise before the stores of the pass that precedes it.
template <int KP, int BM, int WCT, int DEPTH>
__global__ __launch_bounds__(256, 2) void astat_early_kernel(const float* __restrict__ A, int64_t lda, int64_t M, int K,
                                                             const unsigned short* __restrict__ Bp, int N, float* __restrict__ C,
                                                             int64_t ldc, int n_mt) {
    constexpr int PITCH = KP * 2 + 16;
    constexpr int F4R = KP / 4;
    constexpr int ITERS = BM * F4R / 256;
    constexpr int IPC = ITERS / 2;
    constexpr int MR = BM / 16;
    constexpr int NK = KP / 32;
    constexpr int D1 = DEPTH + 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char As[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int K4 = (K + 3) & ~3;
    const uint32_t ld4 = (uint32_t)lda * 4u;
    const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(Bp), 0, 640 * KP * 2, 0x00020000);
    auto tile_rsrc = [&](int mt) {
        const int64_t m0 = (int64_t)mt * BM;
        const int64_t rows_left = mt < n_mt ? M - m0 : 0;
        const uint64_t base = reinterpret_cast<uint64_t>(A + (mt < n_mt ? m0 : 0) * lda);
        const uint64_t bu = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(base >> 32)) << 32) |
                            (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)base);
        const int64_t nbytes = (rows_left < BM ? rows_left : BM) * lda * 4;
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(bu), 0, __builtin_amdgcn_readfirstlane((int)nbytes), 0x00020000);
    };
    auto load_chunk = [&](float4 (&v)[IPC], __amdgpu_buffer_rsrc_t rs, int ch) {
        int tt = tid;
        asm volatile("" : "+v"(tt));
#pragma unroll
        for (int i = 0; i < IPC; ++i) {
            const int idx = tt + 256 * (ch * IPC + i);
            const int r = idx / F4R, c = idx - r * F4R;
            const uint32_t off = c * 4 < K4 ? (uint32_t)r * ld4 + (uint32_t)c * 16u : 0x80000000u;
            const f32x4 t = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0));
            v[i] = make_float4(t.x, t.y, t.z, t.w);
        }
    };
    auto store_chunk = [&](const float4 (&v)[IPC], int ch) {
        int tt = tid;
        asm volatile("" : "+v"(tt));
#pragma unroll
        for (int i = 0; i < IPC; ++i) {
            const int idx = tt + 256 * (ch * IPC + i);
            const int r = idx / F4R, c = idx - r * F4R;
            uint2 w;
            w.x = bf16_pack(v[i].x, v[i].y);
            w.y = bf16_pack(v[i].z, v[i].w);
            *reinterpret_cast<uint2*>(As + r * PITCH + c * 8) = w;
        }
    };
    float4 v[IPC];
    load_chunk(v, tile_rsrc(blockIdx.x), 0);
    for (int mt = blockIdx.x; mt < n_mt; mt += gridDim.x) {
        const int64_t m0 = (int64_t)mt * BM;
        // chunk 0 of this tile is in flight (or has arrived) in v
        store_chunk(v, 0);
        load_chunk(v, tile_rsrc(mt), 1);
        store_chunk(v, 1);
        __syncthreads();
        const __amdgpu_buffer_rsrc_t nrs = tile_rsrc(mt + gridDim.x);
        bf16x8 ring[D1][WCT];
        auto bload = [&](bf16x8 (&b)[WCT], int tile0, int kt) {
            const int kk = kt < NK ? kt : NK - 1;
#pragma unroll
            for (int j = 0; j < WCT; ++j)
                b[j] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(brs, lane * 16, ((tile0 + j) * NK + kk) * 1024, 0));
        };
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) bload(ring[d], (wid * 2) * WCT, d);
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            const int tile0 = (wid * 2 + ps) * WCT;
            const int ncol0 = tile0 * 16;
            f32x4 acc[MR][WCT];
#pragma unroll
            for (int i = 0; i < MR; ++i)
#pragma unroll
                for (int j = 0; j < WCT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            auto kstep = [&](const bf16x8 (&b)[WCT], int kt) {
                bf16x8 af[MR];
#pragma unroll
                for (int i = 0; i < MR; ++i) af[i] = *reinterpret_cast<const bf16x8*>(As + (i * 16 + li) * PITCH + kt * 64 + lg * 16);
#pragma unroll
                for (int j = 0; j < WCT; ++j)
#pragma unroll
                    for (int i = 0; i < MR; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], af[i], acc[i][j], 0, 0, 0);
            };
#pragma unroll 1
            for (int k0 = 0; k0 < NK; k0 += D1) {
#pragma unroll
                for (int u = 0; u < D1; ++u) {
                    if (k0 + u + DEPTH < NK) bload(ring[(u + DEPTH) % D1], tile0, k0 + u + DEPTH);
                    if (k0 + u < NK) kstep(ring[u], k0 + u);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // what the NEXT phase needs first is requested before this pass's stores
            constexpr int kLast = (NK - 1) % D1;      // (ring slot of the last k-step: the prefetch below restarts at slot 0 ... DEPTH-1)
            (void)kLast;
            if (ps == 0) {
#pragma unroll
                for (int d = 0; d < DEPTH; ++d) bload(ring[d], tile0 + WCT, d);
            } else {
                load_chunk(v, nrs, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < MR; ++i) {
                const int64_t row = m0 + i * 16 + li;
#pragma unroll
                for (int j = 0; j < WCT; ++j) {
                    const int col0 = ncol0 + j * 16 + lg * 4;
                    if (row < M && col0 < N)
                        *reinterpret_cast<float4*>(C + row * ldc + col0) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }
}

static unsigned short rne(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
static float bf(unsigned short h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}


// ---- round 4: LOADER / CONSUMER wave specialisation ------------------------------------------------------------------------------
// One 512-thread block per CU, two LDS buffers of 64 rows (2 x 78.8 KB).  Waves 4..7 are LOADERS: they keep the next-but-one
// tile's 38 float4 per thread in flight in registers (requested before the barrier that ends an iteration), round them to bf16 and
// write them into the buffer the consumers are NOT reading.  Waves 0..3 are CONSUMERS: MFMAs on the current buffer + C stores,
// never waiting on global memory except for the L2-resident B fragments.  One barrier per tile.  The three phases of a tile (A
// load, MFMA, C store) overlap by construction instead of relying on two co-resident blocks drifting apart.
template <int KP, int WCT, int DEPTH, int NCW, int PROBE = 0>
__global__ __launch_bounds__(NCW * 64 + 256, 1) void astat_lc_kernel(const float* __restrict__ A, int64_t lda, int64_t M, int K,
                                                          const unsigned short* __restrict__ Bp, int N, float* __restrict__ C,
                                                          int64_t ldc, int n_mt) {
    constexpr int BM = 64, PITCH = KP * 2 + 16, F4R = KP / 4, MR = BM / 16, NK = KP / 32, D1 = DEPTH + 1;
    constexpr int NL = 256, ITERS = BM * F4R / NL;
    static_assert(BM * F4R % NL == 0, "a tile must divide over the loader threads");
    extern __shared__ __attribute__((aligned(16))) unsigned char As[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);          // scalar: the role branch below is wave-uniform
    const int n_my = blockIdx.x < (unsigned)n_mt ? (n_mt - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    constexpr int PASSES = 8 / NCW;             // 8 wave-passes of WCT column tiles cover the 640 columns
    if (wid >= NCW) {
        const int K4 = (K + 3) & ~3;
        const uint32_t ld4 = (uint32_t)lda * 4u;
        auto tile_rsrc = [&](int i) {          // i-th tile of this block; past the end: an empty descriptor (loads return zeros)
            const int mt = (int)blockIdx.x + i * (int)gridDim.x;
            const bool ok = i < n_my;
            const int64_t m0 = ok ? (int64_t)mt * BM : 0;
            const int64_t rows_left = ok ? M - m0 : 0;
            const uint64_t base = reinterpret_cast<uint64_t>(A + m0 * lda);
            const uint64_t bu = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(base >> 32)) << 32) |
                                (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)base);
            const int64_t nbytes = (rows_left < BM ? rows_left : BM) * lda * 4;
            return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(bu), 0, __builtin_amdgcn_readfirstlane((int)nbytes), 0x00020000);
        };
        // every register slot is recycled on its own: slot q's value (tile i+1) is rounded and written to LDS, then the slot
        // immediately requests its piece of tile i+2 -- between 37 and 38 KB-pieces per thread are in flight at all times
        auto cycle = [&](f32x4 (&v)[ITERS], int buf, int next_tile, bool first) {
            const __amdgpu_buffer_rsrc_t rs = tile_rsrc(next_tile);
            int tt = tid - NCW * 64;
            asm volatile("" : "+v"(tt));
            unsigned char* dst = As + buf * (BM * PITCH);
#pragma unroll
            for (int q = 0; q < ((PROBE & 2) ? 0 : ITERS); ++q) {
                const int idx = tt + NL * q;
                const int r = idx / F4R, c = idx - r * F4R;
                if (!first) {
                    uint2 w;
                    w.x = bf16_pack(v[q][0], v[q][1]);
                    w.y = bf16_pack(v[q][2], v[q][3]);
                    *reinterpret_cast<uint2*>(dst + r * PITCH + c * 8) = w;
                }
                const uint32_t off = c * 4 < K4 ? (uint32_t)r * ld4 + (uint32_t)c * 16u : 0x80000000u;
                v[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0));
            }
        };
        f32x4 v[ITERS];
        cycle(v, 0, 0, true);                 // request tile 0
        cycle(v, 0, 1, false);                // tile 0 -> buffer 0, request tile 1
        __syncthreads();
#pragma unroll 1
        for (int i = 0; i < n_my; ++i) {
            cycle(v, (i + 1) & 1, i + 2, false);      // tile i+1 -> the buffer the consumers are not reading; request tile i+2
            __syncthreads();
        }
    } else {
        const int li = lane & 15, lg = lane >> 4;
        const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(Bp), 0, 640 * KP * 2, 0x00020000);
        __syncthreads();
#pragma unroll 1
        for (int i = 0; i < n_my; ++i) {
            const int64_t m0 = (int64_t)((int)blockIdx.x + i * (int)gridDim.x) * BM;
            const unsigned char* Ab = As + (i & 1) * (BM * PITCH);
#pragma unroll 1
            for (int ps = 0; ps < ((PROBE & 8) ? 0 : PASSES); ++ps) {       // PROBE 8: the consumers only keep the barriers
                const int ncol0 = (wid * PASSES + ps) * WCT * 16;
                f32x4 acc[MR][WCT];
#pragma unroll
                for (int a = 0; a < MR; ++a)
#pragma unroll
                    for (int j = 0; j < WCT; ++j) acc[a][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                auto bload = [&](bf16x8 (&b)[WCT], int kt) {
                    const int kk = (PROBE & 1) ? 0 : (kt < NK ? kt : NK - 1);      // PROBE 1: one (L1-resident) fragment set
                    if ((PROBE & 16) && kt >= DEPTH) return;                        // PROBE 16: no B loads after the prologue
#pragma unroll
                    for (int j = 0; j < WCT; ++j)
                        b[j] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(brs, lane * 16, (((PROBE & 1) ? j : ncol0 / 16 + j) * NK + kk) * 1024, 0));
                };
                auto kstep = [&](const bf16x8 (&b)[WCT], int kt) {
                    bf16x8 af[MR];
#pragma unroll
                    for (int a = 0; a < MR; ++a)          // PROBE 32: one LDS fragment per row tile, whatever the k-step
                        af[a] = *reinterpret_cast<const bf16x8*>(Ab + (a * 16 + li) * PITCH + ((PROBE & 32) ? 0 : kt) * 64 + lg * 16);
                    if (PROBE & 64) {                     // PROBE 64: no MFMAs (the operands are consumed by one add each)
#pragma unroll
                        for (int j = 0; j < WCT; ++j) acc[0][j][0] += __builtin_bit_cast(f32x4, b[j])[0] + __builtin_bit_cast(f32x4, af[j % MR])[1];
                        return;
                    }
#pragma unroll
                    for (int j = 0; j < WCT; ++j)
#pragma unroll
                        for (int a = 0; a < MR; ++a) acc[a][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], af[a], acc[a][j], 0, 0, 0);
                };
                bf16x8 ring[D1][WCT];
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
                for (int a = 0; a < MR; ++a) {
                    const int64_t row = m0 + a * 16 + li;
#pragma unroll
                    for (int j = 0; j < WCT; ++j) {
                        const int col0 = ncol0 + j * 16 + lg * 4;
                        if ((PROBE & 4) ? (acc[a][j][0] == 123.4f) : (row < M && col0 < N))
                            *reinterpret_cast<float4*>(C + row * ldc + col0) = make_float4(acc[a][j][0], acc[a][j][1], acc[a][j][2], acc[a][j][3]);
                    }
                }
            }
            __syncthreads();
        }
    }
}


// ---- round 4, second form: 128 rows per tile -------------------------------------------------------------------------------------
// What the probes above showed: per 64-row tile a CU pulls the WHOLE weight matrix (760 KB in fragment order) from L2 -- five times
// the bytes of A -- and that L2 -> L1 stream, not missing overlap, is what the tile phases wait on.  The only lever is rows per tile:
// 128 rows halve it.  LDS then holds ONE buffer (128 x 1,232 B = 157.7 KB), so the loader waves keep the NEXT tile in registers,
// already rounded to bf16 (76 uint2 per thread), and write it to LDS between two barriers when the consumers are done with the
// current one (~1 us of a ~30 us tile).  Consumers: 8 x 5 tiles of 16 x 16 per wave and pass = 160 accumulator registers, A
// fragments read in two halves of four, B one step ahead.
template <int KP, int PROBE = 0>
__global__ __launch_bounds__(512, 1) void astat_lc128_kernel(const float* __restrict__ A, int64_t lda, int64_t M, int K,
                                                             const unsigned short* __restrict__ Bp, int N, float* __restrict__ C,
                                                             int64_t ldc, int n_mt) {
    constexpr int BM = 128, WCT = 5, PITCH = KP * 2 + 16, F4R = KP / 4, MR = BM / 16, NK = KP / 32;
    constexpr int NL = 256, PIECES = BM * F4R / NL, SLOTS = 8;
    static_assert(BM * F4R % NL == 0, "a tile must divide over the loader threads");
    extern __shared__ __attribute__((aligned(16))) unsigned char As[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_my = blockIdx.x < (unsigned)n_mt ? (n_mt - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    if (wid >= 4) {
        const int K4 = (K + 3) & ~3;
        const uint32_t ld4 = (uint32_t)lda * 4u;
        auto tile_rsrc = [&](int i) {
            const int mt = (int)blockIdx.x + i * (int)gridDim.x;
            const bool ok = i < n_my;
            const int64_t m0 = ok ? (int64_t)mt * BM : 0;
            const int64_t rows_left = ok ? M - m0 : 0;
            const uint64_t base = reinterpret_cast<uint64_t>(A + m0 * lda);
            const uint64_t bu = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(base >> 32)) << 32) |
                                (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)base);
            const int64_t nbytes = (rows_left < BM ? rows_left : BM) * lda * 4;
            return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(bu), 0, __builtin_amdgcn_readfirstlane((int)nbytes), 0x00020000);
        };
        uint2 hold[PIECES];
        auto fetch = [&](int i) {              // tile i: HBM -> fp32 staging slots -> bf16 pairs in `hold`
            const __amdgpu_buffer_rsrc_t rs = tile_rsrc(i);
            int tt = tid - 256;
            asm volatile("" : "+v"(tt));
            f32x4 st[SLOTS];
            auto req = [&](int q) {
                const int idx = tt + NL * q;
                const int r = idx / F4R, c = idx - r * F4R;
                const uint32_t off = c * 4 < K4 ? (uint32_t)r * ld4 + (uint32_t)c * 16u : 0x80000000u;
                return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0));
            };
            if (PROBE & 2) {
#pragma unroll
                for (int q = 0; q < PIECES; ++q) hold[q] = make_uint2(0u, 0u);
                return;
            }
#pragma unroll
            for (int q = 0; q < SLOTS; ++q) st[q] = req(q);
#pragma unroll
            for (int q = 0; q < PIECES; ++q) {
                const f32x4 v = st[q % SLOTS];
                hold[q].x = bf16_pack(v[0], v[1]);
                hold[q].y = bf16_pack(v[2], v[3]);
                if (q + SLOTS < PIECES) st[q % SLOTS] = req(q + SLOTS);
            }
        };
        auto flush = [&]() {                   // `hold` -> LDS
            int tt = tid - 256;
            asm volatile("" : "+v"(tt));
#pragma unroll
            for (int q = 0; q < PIECES; ++q) {
                const int idx = tt + NL * q;
                const int r = idx / F4R, c = idx - r * F4R;
                *reinterpret_cast<uint2*>(As + r * PITCH + c * 8) = hold[q];
            }
        };
        fetch(0);
        flush();
        __syncthreads();                        // B0: tile 0 in LDS
#pragma unroll 1
        for (int i = 0; i < n_my; ++i) {
            fetch(i + 1);                       // while the consumers multiply tile i
            __syncthreads();                    // A: consumers are done with the LDS tile
            flush();
            __syncthreads();                    // B: tile i+1 in LDS
        }
    } else {
        const int li = lane & 15, lg = lane >> 4;
        const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(Bp), 0, 640 * KP * 2, 0x00020000);
        __syncthreads();                        // B0
#pragma unroll 1
        for (int i = 0; i < n_my; ++i) {
            const int64_t m0 = (int64_t)((int)blockIdx.x + i * (int)gridDim.x) * BM;
#pragma unroll 1
            for (int ps = 0; ps < 2; ++ps) {
                const int ncol0 = (wid * 2 + ps) * WCT * 16;
                f32x4 acc[MR][WCT];
#pragma unroll
                for (int a = 0; a < MR; ++a)
#pragma unroll
                    for (int j = 0; j < WCT; ++j) acc[a][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                auto bload = [&](bf16x8 (&b)[WCT], int kt) {
                    const int kk = (PROBE & 1) ? 0 : (kt < NK ? kt : NK - 1);
#pragma unroll
                    for (int j = 0; j < WCT; ++j)
                        b[j] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(brs, lane * 16, (((PROBE & 1) ? j : ncol0 / 16 + j) * NK + kk) * 1024, 0));
                };
                auto kstep = [&](const bf16x8 (&b)[WCT], int kt) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        bf16x8 af[4];
#pragma unroll
                        for (int a = 0; a < 4; ++a) af[a] = *reinterpret_cast<const bf16x8*>(As + ((h * 4 + a) * 16 + li) * PITCH + kt * 64 + lg * 16);
#pragma unroll
                        for (int j = 0; j < WCT; ++j)
#pragma unroll
                            for (int a = 0; a < 4; ++a)
                                acc[h * 4 + a][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], af[a], acc[h * 4 + a][j], 0, 0, 0);
                    }
                };
                bf16x8 ring[2][WCT];
                bload(ring[0], 0);
#pragma unroll 1
                for (int k0 = 0; k0 < NK; k0 += 2) {
                    bload(ring[1], k0 + 1);
                    kstep(ring[0], k0);
                    __builtin_amdgcn_sched_barrier(0);
                    bload(ring[0], k0 + 2);
                    if (k0 + 1 < NK) kstep(ring[1], k0 + 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int a = 0; a < MR; ++a) {
                    const int64_t row = m0 + a * 16 + li;
#pragma unroll
                    for (int j = 0; j < WCT; ++j) {
                        const int col0 = ncol0 + j * 16 + lg * 4;
                        if ((PROBE & 4) ? (acc[a][j][0] == 123.4f) : (row < M && col0 < N))
                            *reinterpret_cast<float4*>(C + row * ldc + col0) = make_float4(acc[a][j][0], acc[a][j][1], acc[a][j][2], acc[a][j][3]);
                    }
                }
            }
            __syncthreads();                    // A
            __syncthreads();                    // B
        }
    }
}

template <int KP, int BM, int WCT, int PASSES, int DEPTH, int FR, int PROBE = 0>
static void run(const char* name, const float* dA, int64_t lda, int64_t M, int K, const unsigned short* dB, int N, float* dC, int64_t ldc,
                const std::vector<float>& hA, const std::vector<unsigned short>& hB, int grid) {
    constexpr int PITCH = KP * 2 + 16;
    const int n_mt = (int)((M + BM - 1) / BM);
    const size_t lds = (size_t)BM * PITCH;
    auto kern = astat_kernel<KP, BM, WCT, PASSES, DEPTH, FR, PROBE>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipMemset(dC, 0, (size_t)M * ldc * 4));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, dA, lda, M, K, dB, N, dC, ldc, n_mt);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int reps = 20;
    CK(hipEventRecord(e0));
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, dA, lda, M, K, dB, N, dC, ldc, n_mt);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    // check a few rows against the same arithmetic in double
    double worst = 0;
    const int64_t rows[] = {0, 1, 63, 64, 12345, M - 65, M - 1};
    std::vector<float> hc(N);
    for (int64_t r : rows) {
        CK(hipMemcpy(hc.data(), dC + r * ldc, (size_t)N * 4, hipMemcpyDeviceToHost));
        for (int n = 0; n < N; ++n) {
            double s = 0;
            for (int k = 0; k < K; ++k) s += (double)bf(rne(hA[r * lda + k])) * (double)bf(hB[(size_t)n * KP + k]);
            worst = fmax(worst, fabs(s - hc[n]) / (1e-3 + fabs(s)));
        }
    }
    const double flops = 2.0 * M * N * K, bytes = 4.0 * M * (K + N);
    printf("%-34s grid %4d  lds %6zu B  %.3f ms  %.0f TF  %.2f TB/s of A + C   max rel err %.2e\n", name, grid, lds, ms, flops / ms / 1e9,
           bytes / ms / 1e9, worst);
}

template <int KP, int BM, int WCT, int DEPTH, int PROBE>
static void run_pipe(const char* name, const float* dA, int64_t lda, int64_t M, int K, const unsigned short* dB, int N, float* dC, int64_t ldc,
                const std::vector<float>& hA, const std::vector<unsigned short>& hB, int grid) {
    constexpr int PITCH = KP * 2 + 16;
    const int n_mt = (int)((M + BM - 1) / BM);
    const size_t lds = (size_t)BM * PITCH;
    auto kern = astat_pipe_kernel<KP, BM, WCT, DEPTH, PROBE>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipMemset(dC, 0, (size_t)M * ldc * 4));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, dA, lda, M, K, dB, N, dC, ldc, n_mt);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int reps = 20;
    CK(hipEventRecord(e0));
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, dA, lda, M, K, dB, N, dC, ldc, n_mt);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    // check a few rows against the same arithmetic in double
    double worst = 0;
    const int64_t rows[] = {0, 1, 63, 64, 12345, M - 65, M - 1};
    std::vector<float> hc(N);
    for (int64_t r : rows) {
        CK(hipMemcpy(hc.data(), dC + r * ldc, (size_t)N * 4, hipMemcpyDeviceToHost));
        for (int n = 0; n < N; ++n) {
            double s = 0;
            for (int k = 0; k < K; ++k) s += (double)bf(rne(hA[r * lda + k])) * (double)bf(hB[(size_t)n * KP + k]);
            worst = fmax(worst, fabs(s - hc[n]) / (1e-3 + fabs(s)));
        }
    }
    const double flops = 2.0 * M * N * K, bytes = 4.0 * M * (K + N);
    printf("%-34s grid %4d  lds %6zu B  %.3f ms  %.0f TF  %.2f TB/s of A + C   max rel err %.2e\n", name, grid, lds, ms, flops / ms / 1e9,
           bytes / ms / 1e9, worst);
}

template <int KP, int BM, int WCT, int DEPTH>
static void run_early(const char* name, const float* dA, int64_t lda, int64_t M, int K, const unsigned short* dB, int N, float* dC,
                      int64_t ldc, const std::vector<float>& hA, const std::vector<unsigned short>& hB, int grid) {
    constexpr int PITCH = KP * 2 + 16;
    const int n_mt = (int)((M + BM - 1) / BM);
    const size_t lds = (size_t)BM * PITCH;
    auto kern = astat_early_kernel<KP, BM, WCT, DEPTH>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipMemset(dC, 0, (size_t)M * ldc * 4));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, dA, lda, M, K, dB, N, dC, ldc, n_mt);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int reps = 20;
    CK(hipEventRecord(e0));
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, dA, lda, M, K, dB, N, dC, ldc, n_mt);
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
            double sacc = 0;
            for (int k = 0; k < K; ++k) sacc += (double)bf(rne(hA[r * lda + k])) * (double)bf(hB[(size_t)n * KP + k]);
            worst = fmax(worst, fabs(sacc - hc[n]) / (1e-3 + fabs(sacc)));
        }
    }
    const double flops = 2.0 * M * N * K, bytes = 4.0 * M * (K + N);
    printf("%-34s grid %4d  lds %6zu B  %.3f ms  %.0f TF  %.2f TB/s of A + C   max rel err %.2e\n", name, grid, lds, ms, flops / ms / 1e9,
           bytes / ms / 1e9, worst);
}

template <int KP, int WCT, int DEPTH, int NCW, int PROBE>
static void run_lc(const char* name, const float* dA, int64_t lda, int64_t M, int K, const unsigned short* dB, int N, float* dC, int64_t ldc,
                   const std::vector<float>& hA, const std::vector<unsigned short>& hB, int grid) {
    constexpr int PITCH = KP * 2 + 16, BM = 64;
    const int n_mt = (int)((M + BM - 1) / BM);
    const size_t lds = (size_t)2 * BM * PITCH;
    auto kern = astat_lc_kernel<KP, WCT, DEPTH, NCW, PROBE>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipMemset(dC, 0, (size_t)M * ldc * 4));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(NCW * 64 + 256), lds, 0, dA, lda, M, K, dB, N, dC, ldc, n_mt);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int reps = 20;
    CK(hipEventRecord(e0));
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(NCW * 64 + 256), lds, 0, dA, lda, M, K, dB, N, dC, ldc, n_mt);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    double worst = 0;
    const int64_t rows[] = {0, 1, 63, 64, 12345, 16384 + 77, 64 * 256 * 13 + 5, M - 65, M - 1};
    std::vector<float> hc(N);
    for (int64_t r : rows) {
        CK(hipMemcpy(hc.data(), dC + r * ldc, (size_t)N * 4, hipMemcpyDeviceToHost));
        for (int n = 0; n < N; ++n) {
            double sacc = 0;
            for (int k = 0; k < K; ++k) sacc += (double)bf(rne(hA[r * lda + k])) * (double)bf(hB[(size_t)n * KP + k]);
            worst = fmax(worst, fabs(sacc - hc[n]) / (1e-3 + fabs(sacc)));
        }
    }
    const double flops = 2.0 * M * N * K, bytes = 4.0 * M * (K + N);
    printf("%-34s grid %4d  lds %6zu B  %.3f ms  %.0f TF  %.2f TB/s of A + C   max rel err %.2e\n", name, grid, lds, ms, flops / ms / 1e9,
           bytes / ms / 1e9, worst);
}

template <int KP, int PROBE>
static void run_lc128(const char* name, const float* dA, int64_t lda, int64_t M, int K, const unsigned short* dB, int N, float* dC, int64_t ldc,
                      const std::vector<float>& hA, const std::vector<unsigned short>& hB, int grid) {
    constexpr int PITCH = KP * 2 + 16, BM = 128;
    const int n_mt = (int)((M + BM - 1) / BM);
    const size_t lds = (size_t)BM * PITCH;
    auto kern = astat_lc128_kernel<KP, PROBE>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipMemset(dC, 0, (size_t)M * ldc * 4));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, 0, dA, lda, M, K, dB, N, dC, ldc, n_mt);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int reps = 20;
    CK(hipEventRecord(e0));
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, 0, dA, lda, M, K, dB, N, dC, ldc, n_mt);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    double worst = 0;
    const int64_t rows[] = {0, 1, 63, 64, 127, 128, 12345, 16384 + 77, 128 * 256 * 7 + 5, M - 129, M - 65, M - 1};
    std::vector<float> hc(N);
    for (int64_t r : rows) {
        CK(hipMemcpy(hc.data(), dC + r * ldc, (size_t)N * 4, hipMemcpyDeviceToHost));
        for (int n = 0; n < N; ++n) {
            double sacc = 0;
            for (int k = 0; k < K; ++k) sacc += (double)bf(rne(hA[r * lda + k])) * (double)bf(hB[(size_t)n * KP + k]);
            worst = fmax(worst, fabs(sacc - hc[n]) / (1e-3 + fabs(sacc)));
        }
    }
    const double flops = 2.0 * M * N * K, bytes = 4.0 * M * (K + N);
    printf("%-34s grid %4d  lds %6zu B  %.3f ms  %.0f TF  %.2f TB/s of A + C   max rel err %.2e\n", name, grid, lds, ms, flops / ms / 1e9,
           bytes / ms / 1e9, worst);
}

int main() {
    const int64_t M = 440000;
    const int K = 600, N = 600, KP = 608, NP = 640;
    const int64_t lda = 600, ldc = 600;
    std::vector<float> hA((size_t)M * lda);
    uint32_t s = 12345;
    auto rnd = [&]() {
        s = s * 1664525u + 1013904223u;
        return (float)((s >> 8) & 0xffff) / 65536.f - 0.5f;
    };
    for (auto& x : hA) x = rnd();
    std::vector<unsigned short> hB((size_t)NP * KP, 0);
    for (int n = 0; n < N; ++n)
        for (int k = 0; k < K; ++k) hB[(size_t)n * KP + k] = rne(rnd() * 0.1f);
    float *dA, *dC;
    unsigned short* dB;
    CK(hipMalloc(&dA, hA.size() * 4));
    CK(hipMalloc(&dC, (size_t)M * ldc * 4));
    CK(hipMalloc(&dB, hB.size() * 2));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice));
    // fragment order: [column tile][k-step][lane][8 bf16] -- a wave's fragment load is 1 KB of consecutive bytes (8 full lines)
    std::vector<unsigned short> hF((size_t)NP * KP, 0);
    for (int nt = 0; nt < NP / 16; ++nt)
        for (int kt = 0; kt < KP / 32; ++kt)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 8; ++e)
                    hF[(((size_t)nt * (KP / 32) + kt) * 64 + lane) * 8 + e] = hB[(size_t)(nt * 16 + (lane & 15)) * KP + kt * 32 + (lane >> 4) * 8 + e];
    unsigned short* dF;
    CK(hipMalloc(&dF, hF.size() * 2));
    CK(hipMemcpy(dF, hF.data(), hF.size() * 2, hipMemcpyHostToDevice));
    for (int grid : {512}) {
        run<608, 64, 5, 2, 2, 1>("BM 64, 2x5, B fragment order, 2 ahead", dA, lda, M, K, dF, N, dC, ldc, hA, hB, grid);
        run_lc<608, 5, 2, 4, 0>("LC 4 consumers x 2 passes, B 2 ahead", dA, lda, M, K, dF, N, dC, ldc, hA, hB, 256);
        run_lc<608, 5, 2, 4, 8>("probe: loaders alone", dA, lda, M, K, dF, N, dC, ldc, hA, hB, 256);
        run_lc<608, 5, 2, 4, 4>("no C stores (MFMA + LDS + B from L2)", dA, lda, M, K, dF, N, dC, ldc, hA, hB, 256);
        run_lc<608, 5, 2, 4, 4 + 1>("  ... B L1-resident", dA, lda, M, K, dF, N, dC, ldc, hA, hB, 256);
        run_lc<608, 5, 2, 4, 4 + 16>("  ... no B loads", dA, lda, M, K, dF, N, dC, ldc, hA, hB, 256);
        run_lc<608, 5, 2, 4, 4 + 16 + 32>("  ... no B loads, A frags cached", dA, lda, M, K, dF, N, dC, ldc, hA, hB, 256);
        run_lc<608, 5, 2, 4, 4 + 64>("  ... B from L2 + LDS reads, no MFMA", dA, lda, M, K, dF, N, dC, ldc, hA, hB, 256);
        run_lc<608, 5, 2, 4, 4 + 64 + 32>("  ... B from L2 only, no MFMA", dA, lda, M, K, dF, N, dC, ldc, hA, hB, 256);
        run_lc<608, 5, 2, 4, 2 + 4>("consumers alone: MFMA + LDS + B from L2", dA, lda, M, K, dF, N, dC, ldc, hA, hB, 256);
        run_lc<608, 5, 2, 4, 2 + 4 + 64>("consumers alone: B from L2 + LDS, no MFMA", dA, lda, M, K, dF, N, dC, ldc, hA, hB, 256);
        run_lc<608, 5, 2, 4, 2 + 4 + 16>("consumers alone: MFMA + LDS, no B loads", dA, lda, M, K, dF, N, dC, ldc, hA, hB, 256);
        run<608, 64, 5, 2, 2, 1>("BM 64, 2x5, B fragment order, 2 ahead", dA, lda, M, K, dF, N, dC, ldc, hA, hB, grid);
    }
    {
        int nb = 0;
        auto kern = astat_kernel<608, 64, 5, 2, 2, 1, 0>;
        CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * (608 * 2 + 16)));
        CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, 256, 64 * (608 * 2 + 16)));
        printf("occupancy reported by the runtime: %d blocks per CU\n", nb);
    }
    return 0;
}
