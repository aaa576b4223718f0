// Micro-benchmark (NOT part of the product; round 6): joules per bf16 MFMA flop of the two instruction shapes the split-bf16 kernels
// could be built on -- v_mfma_f32_16x16x32_bf16 (what gemm_x3.hip uses: 4 x 5 tiles of 16 x 16 per wave = 80 accumulator registers) and
// v_mfma_f32_32x32x16_bf16 (1 x 5 tiles of 32 x 32 = 80 registers) -- with operands in registers only, two waves per SIMD, equal flops.
// The whole-rows kernel is bound by the socket's 1.4 kW cap (profiles/r05_x3_energy.txt): if the 32 x 32 shape spends fewer joules per
// flop, a kernel built on it runs at a higher clock.  Run under tools/clock_watch.py:  mfma_energy <0|1|2|3> [launches]
//   0: 16x16x32, 4 x 5 tiles, A / B fragments constant      1: 32x32x16, 1 x 5 tiles, constant fragments
//   2 / 3: the same with the fragments re-read from LDS every k-step the way the kernel's loop does (12 + 0 / 3 + 0 ds_read_b128 per 32 k:
//          the B fragments stay in registers, as they come from L2 in the kernel)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int KSTEPS = 2048;          // 32-k steps per wave

template <bool LDS>
__global__ __launch_bounds__(256, 2) void k16(float* out, const bf16x8* seed) {
    __shared__ bf16x8 sm[3 * 64 * 4];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 3 * 64 * 4; i += 256) sm[i] = seed[i & 63];
    __syncthreads();
    f32x4 acc[4][5];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 af[4][3], bf[5][3];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int p = 0; p < 3; ++p) af[i][p] = seed[(lane + i + p) & 63];
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int p = 0; p < 3; ++p) bf[j][p] = seed[(lane + 7 * j + p) & 63];
#pragma unroll 1
    for (int s = 0; s < KSTEPS; ++s) {
        if (LDS) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int p = 0; p < 3; ++p) af[i][p] = sm[(p * 4 + i) * 64 + ((lane + s) & 63)];
        }
#pragma unroll
        for (int j = 0; j < 5; ++j) {
#define T(PB, PA) _Pragma("unroll") for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j][PB], af[i][PA], acc[i][j], 0, 0, 0);
            T(0, 2) T(2, 0) T(1, 1) T(0, 1) T(1, 0) T(0, 0)
#undef T
        }
    }
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j) r += acc[i][j][0] + acc[i][j][3];
    if (r == 123.456f) out[threadIdx.x] = r;
}

// 32 x 32 x 16: a wave owns 64 rows (two row tiles) x ... the same 80 accumulator registers as above are 1 row tile x 5 column tiles of
// 32 x 32; per 32 k (two k-steps of 16) the 64-row block needs, per wave, 2 k-halves x 3 planes of ONE row tile = 6 A fragments
template <bool LDS>
__global__ __launch_bounds__(256, 2) void k32(float* out, const bf16x8* seed) {
    __shared__ bf16x8 sm[3 * 64 * 4];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 3 * 64 * 4; i += 256) sm[i] = seed[i & 63];
    __syncthreads();
    f32x16 acc[5];
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;
    bf16x8 af[2][3], bf[5][2][3];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int p = 0; p < 3; ++p) af[h][p] = seed[(lane + h + p) & 63];
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int p = 0; p < 3; ++p) bf[j][h][p] = seed[(lane + 7 * j + 3 * h + p) & 63];
    // equal flops: a 16x16x32 wave above issues 120 MFMAs of 16,384 flops per 32-k step; here 60 MFMAs of 32,768
#pragma unroll 1
    for (int s = 0; s < KSTEPS; ++s) {
        if (LDS) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int p = 0; p < 3; ++p) af[h][p] = sm[(p * 4 + h) * 64 + ((lane + s) & 63)];
        }
#pragma unroll
            for (int j = 0; j < 5; ++j) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
#define T(PB, PA) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[j][h][PB], af[h][PA], acc[j], 0, 0, 0);
                    T(0, 2) T(2, 0) T(1, 1) T(0, 1) T(1, 0) T(0, 0)
#undef T
                }
            }
    }
    float r = 0.f;
#pragma unroll
    for (int j = 0; j < 5; ++j) r += acc[j][0] + acc[j][15];
    if (r == 123.456f) out[threadIdx.x] = r;
}

int main(int argc, char** argv) {
    const int which = argc > 1 ? atoi(argv[1]) : 0;
    const int launches = argc > 2 ? atoi(argv[2]) : 2000;
    float* out;
    bf16x8* seed;
    CK(hipMalloc(&out, 4096));
    CK(hipMalloc(&seed, 64 * sizeof(bf16x8)));
    short h[64 * 8];
    for (int i = 0; i < 64 * 8; ++i) h[i] = (short)(0x3c00 + (i * 37) % 1024);          // bf16 values around 0.01 .. 0.03
    CK(hipMemcpy(seed, h, sizeof(h), hipMemcpyHostToDevice));
    void (*kern)(float*, const bf16x8*) = which == 0 ? k16<false> : which == 1 ? k32<false> : which == 2 ? k16<true> : k32<true>;
    const char* names[] = {"16x16x32, 4 x 5 tiles, registers only", "32x32x16, 1 x 5 tiles, registers only", "16x16x32 + A fragments from LDS (12 ds_read_b128 per 32 k)",
                           "32x32x16 + A fragments from LDS (6 ds_read_b128 per 32 k)"};
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kern, dim3(512), dim3(256), 0, 0, out, seed);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    for (int w = 0; w < launches; ++w) hipLaunchKernelGGL(kern, dim3(512), dim3(256), 0, 0, out, seed);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= launches;
    const double flops = 512.0 * 4 * KSTEPS * 120 * 16384.0;          // per launch
    printf("[%d] %-62s %.3f ms per launch  %.0f TF bf16\n", which, names[which], ms, flops / ms / 1e9);
    return 0;
}
