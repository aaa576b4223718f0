// Micro-benchmark (NOT part of the product): would v_mfma_f32_32x32x2_f32 serve the N = 300 contractions better than
// v_mfma_f32_16x16x4_f32?  Both are exact fp32 fma chains with the same FLOP per pipe cycle (4096 flop / 64 cycles vs
// 2048 / 32); the 32x32 form issues half the instructions and reads half the A/B operand registers per flop, but a wave
// tile has to be a multiple of 32 in both directions.  For N = 300 -> 320 = 2 x 160 and 160 = 5 x 32, the only 4-wave
// split of a 128 x 160 block is 4 x 1 waves of 32 x 160 (1 x 5 tiles: 24 LDS fragment float4s per stage and lane against
// 18 for the 64 x 80 tile of the 16x16 form).  Loop shapes:
//   A  16x16x4, wave tile 64 x 80 (gemm.hip today)      B  32x32x2, wave tile 32 x 160      C  32x32x2, wave tile 64 x 64 (2 x 2:
//   what a 128 x 128 block would use -- N = 256 shapes)
// each: registers only / + LDS fragment reads + barrier + LDS stores per stage, 2 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_32x32.hip -o gpurun_out/mfma_32x32 && gpurun_out/mfma_32x32
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int KP = 36;          // k-contiguous LDS pitch (floats)

// SHAPE 0: 16x16x4 64x80;  1: 32x32x2 32x160;  2: 32x32x2 64x64.   FULL: 0 registers only, 1 = LDS reads + stores + barrier
template <int SHAPE, int FULL>
__global__ __launch_bounds__(256, 2) void loop_kernel(int stages, float* out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int e = threadIdx.x; e < 2 * 288 * KP; e += 256) smem[e] = (float)(e & 7) * 0.125f;
    __syncthreads();
    const float* As = smem;                 // 128 rows x 36
    const float* Bs = smem + 128 * KP;      // 160 rows x 36 (both operands k-contiguous: the A . B^T case)
    float4 st[9];
    for (int i = 0; i < 9; ++i) st[i] = make_float4(1.f + i, 2.f, 3.f, 4.f);
    float r = 0.f;
    if constexpr (SHAPE == 0) {
        constexpr int MR = 4, NR = 5;
        const int wm = wid >> 1, wn = wid & 1, li = lane & 15, lg = lane >> 4;
        f32x4 acc[MR][NR];
        for (int i = 0; i < MR; ++i)
            for (int j = 0; j < NR; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        float4 af[MR], bf[NR];
        for (int i = 0; i < MR; ++i) af[i] = make_float4(0.5f + lane * 1e-3f, 1.f, 2.f, 3.f);
        for (int j = 0; j < NR; ++j) bf[j] = make_float4(0.25f + lane * 1e-3f, 1.f, 2.f, 3.f);
        for (int s = 0; s < stages; ++s) {
#pragma unroll
            for (int kk = 0; kk < 32; kk += 16) {
                if (FULL) {
#pragma unroll
                    for (int i = 0; i < MR; ++i) af[i] = *reinterpret_cast<const float4*>(As + (wm * 64 + i * 16 + li) * KP + kk + 4 * lg);
#pragma unroll
                    for (int j = 0; j < NR; ++j) bf[j] = *reinterpret_cast<const float4*>(Bs + (wn * 80 + j * 16 + li) * KP + kk + 4 * lg);
                }
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int i = 0; i < MR; ++i)
#pragma unroll
                        for (int j = 0; j < NR; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(((const float*)&bf[j])[t], ((const float*)&af[i])[t], acc[i][j], 0, 0, 0);
            }
            if (FULL) {
                float* dst = smem + 288 * KP;
#pragma unroll
                for (int i = 0; i < 9; ++i) *reinterpret_cast<float4*>(dst + (threadIdx.x + 256 * i) * 4) = st[i];
                __syncthreads();
            }
        }
        for (int i = 0; i < MR; ++i)
            for (int j = 0; j < NR; ++j) r += acc[i][j][0] + acc[i][j][3];
    } else {
        constexpr int MR = SHAPE == 1 ? 1 : 2, NR = SHAPE == 1 ? 5 : 2;
        const int wm = SHAPE == 1 ? wid : (wid >> 1), wn = SHAPE == 1 ? 0 : (wid & 1);
        const int l32 = lane & 31, h = lane >> 5;
        f32x16 acc[MR][NR];
        for (int i = 0; i < MR; ++i)
            for (int j = 0; j < NR; ++j)
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        float4 af[MR][4], bf[NR][4];
        for (int i = 0; i < MR; ++i)
            for (int q = 0; q < 4; ++q) af[i][q] = make_float4(0.5f + lane * 1e-3f, 1.f, 2.f, 3.f);
        for (int j = 0; j < NR; ++j)
            for (int q = 0; q < 4; ++q) bf[j][q] = make_float4(0.25f + lane * 1e-3f, 1.f, 2.f, 3.f);
        for (int s = 0; s < stages; ++s) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                // half h of the wave reads the float4 at k = 8q + 4h: MFMA (q, c) multiplies k = 8q + c (half 0) and 8q + 4 + c
                if (FULL) {
#pragma unroll
                    for (int i = 0; i < MR; ++i) af[i][q] = *reinterpret_cast<const float4*>(As + (wm * 32 * MR + i * 32 + l32) * KP + 8 * q + 4 * h);
#pragma unroll
                    for (int j = 0; j < NR; ++j) bf[j][q] = *reinterpret_cast<const float4*>(Bs + (wn * 32 * NR + j * 32 + l32) * KP + 8 * q + 4 * h);
                }
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int i = 0; i < MR; ++i)
#pragma unroll
                        for (int j = 0; j < NR; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(((const float*)&bf[j][q])[c], ((const float*)&af[i][q])[c], acc[i][j], 0, 0, 0);
            }
            if (FULL) {
                float* dst = smem + 288 * KP;
#pragma unroll
                for (int i = 0; i < 9; ++i) *reinterpret_cast<float4*>(dst + (threadIdx.x + 256 * i) * 4) = st[i];
                __syncthreads();
            }
        }
        for (int i = 0; i < MR; ++i)
            for (int j = 0; j < NR; ++j) r += acc[i][j][0] + acc[i][j][15];
    }
    if (r == 123.456f) out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int SHAPE, int FULL>
void run(const char* name, float* out, double flop_per_wave_stage) {
    const int stages = 4000;
    const size_t lds = 2 * 288 * KP * sizeof(float);
    hipFuncSetAttribute((const void*)loop_kernel<SHAPE, FULL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int grid = 512;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int w = 0; w < 3; ++w) loop_kernel<SHAPE, FULL><<<grid, 256, lds>>>(stages, out);
    hipDeviceSynchronize();
    float best = 1e30f, sum = 0.f;
    const int reps = 8;
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(a);
        loop_kernel<SHAPE, FULL><<<grid, 256, lds>>>(stages, out);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        best = ms < best ? ms : best;
        sum += ms;
    }
    const double flops = (double)grid * 4 * stages * flop_per_wave_stage;
    printf("%-64s best %.3f ms  %.1f TFLOP/s (mean %.1f)\n", name, best, flops / best / 1e9, flops / (sum / reps) / 1e9);
    if (hipGetLastError() != hipSuccess) printf("  launch error\n");
}

int main() {
    float* out;
    hipMalloc(&out, 1 << 22);
    run<0, 0>("16x16x4  wave 64x80   registers only", out, 64.0 * 80 * 32 * 2);
    run<1, 0>("32x32x2  wave 32x160  registers only", out, 32.0 * 160 * 32 * 2);
    run<2, 0>("32x32x2  wave 64x64   registers only", out, 64.0 * 64 * 32 * 2);
    run<0, 1>("16x16x4  wave 64x80   + LDS reads, stores, barrier", out, 64.0 * 80 * 32 * 2);
    run<1, 1>("32x32x2  wave 32x160  + LDS reads, stores, barrier", out, 32.0 * 160 * 32 * 2);
    run<2, 1>("32x32x2  wave 64x64   + LDS reads, stores, barrier", out, 64.0 * 64 * 32 * 2);
    return 0;
}
