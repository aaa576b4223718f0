// Shared host-side helpers for libgeogcn.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/geogcn.h"

namespace geogcn {

constexpr int kWave = 64;          // CDNA4 wavefront
constexpr int kNumCU = 256;        // MI355X
constexpr int kNumXCD = 8;

void set_error(const char* fmt, ...);

#define GEOGCN_REQUIRE(cond, code, ...)             \
    do {                                            \
        if (!(cond)) {                              \
            geogcn::set_error(__VA_ARGS__);         \
            return (code);                          \
        }                                           \
    } while (0)

// launch check: kernel launch errors surface through hipGetLastError (no sync)
#define GEOGCN_LAUNCH_CHECK(name)                                              \
    do {                                                                       \
        hipError_t e__ = hipGetLastError();                                    \
        if (e__ != hipSuccess) {                                               \
            geogcn::set_error("%s: %s", name, hipGetErrorString(e__));         \
            return (int)e__;                                                   \
        }                                                                      \
    } while (0)

#define GEOGCN_HIP(call)                                                       \
    do {                                                                       \
        hipError_t e__ = (call);                                               \
        if (e__ != hipSuccess) {                                               \
            geogcn::set_error("%s: %s", #call, hipGetErrorString(e__));        \
            return (int)e__;                                                   \
        }                                                                      \
    } while (0)

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// device-side activation shared by SpMM / GEMM / elementwise epilogues
template <int ACT>
__device__ __forceinline__ float apply_act(float x) {
    if constexpr (ACT == GEOGCN_ACT_TANH) {
        return tanhf(x);
    } else if constexpr (ACT == GEOGCN_ACT_SIGMOID) {
        return 1.0f / (1.0f + expf(-x));
    } else if constexpr (ACT == GEOGCN_ACT_SELU) {          // scale * elu(x, alpha) (Klambauer et al. 2017)
        return 1.0507009873554805f * (x > 0.f ? x : 1.6732632423543772f * (expf(x) - 1.0f));
    } else if constexpr (ACT == GEOGCN_ACT_RELU) {
        return x > 0.f ? x : 0.f;
    } else {
        return x;
    }
}

// gemm_bf16.hip
size_t gemm_bf16_workspace_bytes(int precision, int64_t N, int64_t K);
int gemm_bf16_dispatch(int precision, int transB, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
                       const float* B, int64_t ldb, float* C, int64_t ldc, const float* bias, int act, int accumulate,
                       void* ws, size_t ws_bytes, hipStream_t st);

}  // namespace geogcn
