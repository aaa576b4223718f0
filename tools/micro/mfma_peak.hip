// Micro-benchmark (NOT part of the product): what fp32 MFMA rate does an MI355X sustain, and what do the
// ingredients of a GEMM main loop (LDS fragment reads, per-stage barrier, LDS stores) cost on top?
// Shape of the inner loop = gemm.hip's NN <128,160> kernel: per stage 2 x (4 A + 5 B fragments of 4 k-steps,
// 80 MFMAs), wave tile 64 x 80 = 20 accumulators, 4 waves per block, WPS waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_peak.hip -o gpurun_out/mfma_peak && gpurun_out/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int MR = 4, NR = 5, KP = 36, BP = 164;

// MODE 0: registers only;  1: + LDS fragment reads;  2: + barrier per stage;  3: + LDS stores of a stage
template <int MODE, int WPS>
__global__ __launch_bounds__(256, WPS) void loop_kernel(int stages, float* out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int wm = wid >> 1, wn = wid & 1, li = lane & 15, lg = lane >> 4;
    f32x4 acc[MR][NR];
    for (int i = 0; i < MR; ++i)
        for (int j = 0; j < NR; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float* As = smem;
    float* Bs = smem + 128 * KP;
    for (int e = threadIdx.x; e < 128 * KP + 32 * BP; e += 256) smem[e] = (float)(e & 7) * 0.125f;
    __syncthreads();
    float af[MR][4], bf[NR][4];
    for (int i = 0; i < MR; ++i)
        for (int t = 0; t < 4; ++t) af[i][t] = 0.5f + lane * 1e-3f + t;
    for (int j = 0; j < NR; ++j)
        for (int t = 0; t < 4; ++t) bf[j][t] = 0.25f + lane * 1e-3f + t;
    float4 st[9];
    for (int i = 0; i < 9; ++i) st[i] = make_float4(1.f + i, 2.f, 3.f, 4.f);
    for (int s = 0; s < stages; ++s) {
#pragma unroll
        for (int kk = 0; kk < 32; kk += 16) {
            if (MODE >= 1) {
#pragma unroll
                for (int i = 0; i < MR; ++i) {
                    const float4 v = *reinterpret_cast<const float4*>(As + (wm * 64 + i * 16 + li) * KP + kk + 4 * lg);
                    af[i][0] = v.x; af[i][1] = v.y; af[i][2] = v.z; af[i][3] = v.w;
                }
#pragma unroll
                for (int j = 0; j < NR; ++j)
#pragma unroll
                    for (int t = 0; t < 4; ++t) bf[j][t] = Bs[(kk + 4 * lg + t) * BP + wn * 80 + j * 16 + li];
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < MR; ++i)
#pragma unroll
                    for (int j = 0; j < NR; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][t], bf[j][t], acc[i][j], 0, 0, 0);
        }
        if (MODE >= 3) {
            // a stage's worth of LDS stores (9 float4 per thread) into the second half of the buffer
            float* dst = smem + 128 * KP + 32 * BP;
#pragma unroll
            for (int i = 0; i < 9; ++i) *reinterpret_cast<float4*>(dst + (threadIdx.x + 256 * i) * 4) = st[i];
        }
        if (MODE >= 2) __syncthreads();
    }
    float r = 0.f;
    for (int i = 0; i < MR; ++i)
        for (int j = 0; j < NR; ++j) r += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (r == 123.456f) out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int MODE, int WPS>
void run(const char* name, float* out) {
    const int stages = 4000;
    const size_t lds = 2 * (128 * KP + 32 * BP) * sizeof(float);
    hipFuncSetAttribute((const void*)loop_kernel<MODE, WPS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int grid = 256 * WPS;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    loop_kernel<MODE, WPS><<<grid, 256, lds>>>(200, out);
    hipDeviceSynchronize();
    float best = 1e30f, sum = 0.f;
    const int reps = 5;
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(a);
        loop_kernel<MODE, WPS><<<grid, 256, lds>>>(stages, out);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        best = ms < best ? ms : best;
        sum += ms;
    }
    const double flops = (double)grid * 4 * stages * 160 * 2048.0;
    printf("%-44s waves/SIMD %d  best %.3f ms  %.1f TFLOP/s (mean %.1f)\n", name, WPS, best, flops / best / 1e9,
           flops / (sum / reps) / 1e9);
    if (hipGetLastError() != hipSuccess) printf("  launch error\n");
}

int main() {
    float* out;
    hipMalloc(&out, 1 << 22);
    run<0, 1>("registers only", out);
    run<0, 2>("registers only", out);
    run<1, 2>("+ LDS fragment reads", out);
    run<2, 2>("+ barrier per stage", out);
    run<3, 2>("+ LDS stores per stage", out);
    run<3, 1>("+ LDS stores per stage", out);
    // sustained: 20 back-to-back launches of the full loop (power / clock behaviour)
    {
        const size_t lds = 2 * (128 * KP + 32 * BP) * sizeof(float);
        hipEvent_t a, b;
        hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a);
        for (int r = 0; r < 40; ++r) loop_kernel<3, 2><<<512, 256, lds>>>(4000, out);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        printf("sustained 40 launches of the full loop: %.1f TFLOP/s\n", 40.0 * 512 * 4 * 4000 * 160 * 2048.0 / ms / 1e9);
    }
    return 0;
}
