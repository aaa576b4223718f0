// Micro-benchmark (NOT part of the product): where does the fp32 GEMM kernel of libgeogcn lose time?
// Instantiates gemm.hip's kernel with its PROBE switches (parts of the loop removed) on the TwitterUS
// NN shape (440000 x 300) . (300 x 300).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/gemm_variants.hip -o tools/micro/bin/gemm_variants
#include "../../geographconv_amd/csrc/gemm.hip"

#include <stdarg.h>
#include <vector>

namespace geogcn {
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vfprintf(stderr, fmt, ap);
    va_end(ap);
    fputc('\n', stderr);
}
size_t gemm_bf16_workspace_bytes(int, int64_t, int64_t) { return 0; }
size_t gemm_bf16_tn_workspace_bytes(int64_t, int64_t, int64_t) { return 0; }
int gemm_bf16_dispatch(int, int, int64_t, int64_t, int64_t, const float*, int64_t, const float*, int64_t, void*, int64_t, int,
                       const float*, int, int, void*, size_t, hipStream_t) { return -1; }
int gemm_bf16_tn_dispatch(int64_t, int64_t, int64_t, const float*, int64_t, const float*, int64_t, float*, int64_t,
                          const float*, int, int, void*, size_t, hipStream_t) { return 1; }
int zero_fill_async(void*, size_t, hipStream_t) { return 0; }
}  // namespace geogcn

template <int BM, int BN, bool BT, int PROBE>
void run(const char* name, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B, int64_t ldb,
         float* C, int64_t ldc) {
    using Cfg = GemmCfg<BM, BN, false, BT>;
    const SplitPlan sp = plan_grid<BM, BN, false, BT>(M, N, K);
    GemmArgs a{M, N, K, A, lda, B, ldb, C, ldc, nullptr, 0, sp.kchunk, (int)cdiv(M, BM), (int)cdiv(N, BN), sp.nsplit, 1};
    auto kern = gemm_kernel<BM, BN, false, BT, 0, 0, PROBE>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::kLdsBytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(sp.grid), dim3(TPB), Cfg::kLdsBytes, 0, a);
    hipDeviceSynchronize();
    const int reps = 20;
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(sp.grid), dim3(TPB), Cfg::kLdsBytes, 0, a);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    printf("%-58s %.3f ms  %.1f TFLOP/s useful\n", name, ms, 2.0 * M * N * K / ms / 1e9);
    if (hipGetLastError() != hipSuccess) printf("   launch error\n");
}

int main() {
    const int64_t M = 440000, N = 300, K = 300, ld = 320, ldw = 300;
    float *A, *B, *C;
    hipMalloc(&A, M * ld * 4);
    hipMalloc(&B, 320 * 320 * 4);
    hipMalloc(&C, M * ld * 4);
    hipMemset(A, 0, M * ld * 4);
    hipMemset(B, 0, 320 * 320 * 4);
    run<128, 160, false, 0>("NN 128x160 as shipped", M, N, K, A, ld, B, ldw, C, ld);
    run<128, 160, false, 16>("NN  + s_setprio(1) around the MFMA section", M, N, K, A, ld, B, ldw, C, ld);
    run<128, 160, false, 4>("NN  no K-tail early-out", M, N, K, A, ld, B, ldw, C, ld);
    run<128, 160, false, 2>("NN  no C stores", M, N, K, A, ld, B, ldw, C, ld);
    run<128, 160, false, 1>("NN  no global loads", M, N, K, A, ld, B, ldw, C, ld);
    run<128, 160, false, 3>("NN  no global loads, no C stores", M, N, K, A, ld, B, ldw, C, ld);
    run<128, 160, false, 11>("NN  no global loads, no C stores, no LDS stores", M, N, K, A, ld, B, ldw, C, ld);
    run<128, 160, false, 15>("NN  ... and no K-tail early-out", M, N, K, A, ld, B, ldw, C, ld);
    run<96, 160, true, 0>("NT 96x160 as shipped", M, N, K, A, ld, B, ldw, C, ld);
    run<96, 160, true, 16>("NT  + s_setprio(1) around the MFMA section", M, N, K, A, ld, B, ldw, C, ld);
    run<96, 160, true, 2>("NT  no C stores", M, N, K, A, ld, B, ldw, C, ld);
    run<96, 160, true, 1>("NT  no global loads", M, N, K, A, ld, B, ldw, C, ld);
    run<96, 160, true, 11>("NT  no global loads, no C stores, no LDS stores", M, N, K, A, ld, B, ldw, C, ld);
    return 0;
}
