// Shared host-side helpers for libgeogcn.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/geogcn.h"

namespace geogcn {

// x + g * (1 - t) with every operation rounded on its own (no fused multiply-add): the value highway_bwd's hw_grad stores as the
// carry, added the way an accumulating epilogue adds it (gemm.hip / gemm_bf16.hip whole-rows kernels)
__device__ __forceinline__ float add_gate_carry(float x, float g, float t) {
#pragma clang fp contract(off)
    const float c = g * (1.0f - t);
    return x + c;
}

// dropout-masked tanh gradient, the arithmetic of act_bwd (elementwise.hip) spelled out so that every kernel that forms it -- the
// elementwise ones and the whole-rows GEMM's epilogue -- produces the same bits: (g * (keep * scale)) * fma(-y, y, 1)
__device__ __forceinline__ float tanh_bwd_val(float gg, float y) {
#pragma clang fp contract(off)
    return gg * __builtin_fmaf(-y, y, 1.0f);
}
__device__ __forceinline__ float masked_tanh_bwd(float g, float keep, float scale, float y) {
#pragma clang fp contract(off)
    return tanh_bwd_val(g * (keep * scale), y);
}

constexpr int kWave = 64;          // CDNA4 wavefront
constexpr int kNumCU = 256;        // MI355X
constexpr int kNumXCD = 8;

void set_error(const char* fmt, ...);

// The library's two TEST SEAMS (geographconv_amd/tuning.py lists them; nothing else in csrc/ reads the environment): an integer from the
// environment, read at EVERY call so that a test's monkeypatch.setenv / delenv takes effect on the next launch (core.hip).
//   GEOGCN_X3_ROWS_MIN_M     rows from which the split-bf16 whole-rows kernel takes an A . B (gemm_x3.hip; default 4,096)
//   GEOGCN_TN_SLAB_LIMIT     bytes one buffer descriptor is taken to bound in the A^T . B slab kernels (gemm.hip; default 2^31 - 1)
int64_t test_seam_i64(const char* name, int64_t dflt);

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

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel instantiation, DEVICE): the attribute belongs to the device's copy of
// the function, so a process that drives several GPUs sets it on each (a `static bool` per kernel set it on the first device only)
struct LdsAttrOnce {
    uint64_t done = 0;          // bit d: set on device d (devices >= 64: set at every launch)
    int ensure(const void* kern, int bytes) {
        int dev = 0;
        GEOGCN_HIP(hipGetDevice(&dev));
        if (dev < 64 && (done >> dev & 1)) return 0;
        GEOGCN_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        if (dev < 64) done |= (uint64_t)1 << dev;
        return 0;
    }
};

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// device-side activation shared by SpMM / GEMM / elementwise epilogues
template <int ACT>
__device__ __forceinline__ float apply_act(float x) {
    if constexpr (ACT == GEOGCN_ACT_TANH) {
        return tanhf(x);
    } else if constexpr (ACT == GEOGCN_ACT_SIGMOID) {
        // 1 / (1 + exp(-x)) in ~10 instructions (libm's expf plus an IEEE division are ~45, which cost the
        // gate GEMM's epilogue 12 % of its run time): exp(-x) = 2^n * 2^r with n = rint(t), t = -x*log2(e),
        // and r = t - n from a two-term product (|r| <= 0.5, so v_exp_f32's 1 ulp is all that is lost);
        // v_rcp_f32 is 1 ulp.  Measured against fp64: <= 3 ulp of the result.
        const float hi = 1.44269502e+00f, lo = 1.92596299e-08f;      // log2(e) = hi + lo
        const float xc = fminf(fmaxf(x, -88.f), 88.f);              // saturated beyond (also +-inf)
        const float n = rintf(-xc * hi);
        float r = fmaf(-xc, hi, -n);
        r = fmaf(-xc, lo, r);
        const float e = ldexpf(__builtin_amdgcn_exp2f(r), (int)n);
        const float s = __builtin_amdgcn_rcpf(1.0f + e);
        return x != x ? x : s;                                      // fmin/fmax drop a NaN: put it back
    } else if constexpr (ACT == GEOGCN_ACT_SELU) {          // scale * elu(x, alpha) (Klambauer et al. 2017)
        return 1.0507009873554805f * (x > 0.f ? x : 1.6732632423543772f * (expf(x) - 1.0f));
    } else if constexpr (ACT == GEOGCN_ACT_RELU) {
        return x > 0.f ? x : 0.f;
    } else {
        return x;
    }
}

// Four consecutive bias values bias[col0 .. col0 + 3] for an epilogue that owns a float4 of a row: ONE 16-byte load where the quad lies
// inside [0, F) and the vector is 16-byte aligned (every bias of ParamStore's arena is), scalar loads otherwise; beyond F: zeros.
// (Round 6: the epilogues read their bias one float at a time -- 40 scalar loads per lane and row at 600 columns: the 600-wide graph
// product took 2.50 ms with a bias and 2.10 without, tools/spmm_act_probe.py.)  `vec_ok` = aligned16(bias), wave-uniform.
__device__ __forceinline__ float4 load_bias4(const float* __restrict__ bias, int col0, int F, bool vec_ok) {
    if (!bias) return make_float4(0.f, 0.f, 0.f, 0.f);
    if (vec_ok && col0 + 3 < F) return *reinterpret_cast<const float4*>(bias + col0);
    return make_float4(col0 < F ? bias[col0] : 0.f, col0 + 1 < F ? bias[col0 + 1] : 0.f, col0 + 2 < F ? bias[col0 + 2] : 0.f,
                       col0 + 3 < F ? bias[col0 + 3] : 0.f);
}

// ---- Philox4x32-10 (Salmon et al. 2011): the dropout streams (elementwise.hip; fused epilogue of spmm_hot.hip) ----------
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t (&k)[2]) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    const uint32_t hi0 = __umulhi(M0, c[0]), lo0 = M0 * c[0];
    const uint32_t hi1 = __umulhi(M1, c[2]), lo1 = M1 * c[2];
    const uint32_t n0 = hi1 ^ c[1] ^ k[0], n1 = lo1, n2 = hi0 ^ c[3] ^ k[1], n3 = lo0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k[0] += 0x9E3779B9u;
    k[1] += 0xBB67AE85u;
}
// the four 24-bit uniforms [0,1) of counter `ctr` (= element index / 4 + stream offset) under key `seed`
__device__ __forceinline__ void philox_uniform4(uint64_t seed, uint64_t ctr, float (&u)[4]) {
    uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u};
    uint32_t k[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
#pragma unroll
    for (int r = 0; r < 10; ++r) philox_round(c, k);
#pragma unroll
    for (int i = 0; i < 4; ++i) u[i] = (float)(c[i] >> 8) * (1.0f / 16777216.0f);
}

// Zero-fill by a kernel instead of hipMemsetAsync: inside a captured step (hipGraph) a memset node followed
// by a kernel that accumulates into the same buffer was observed to race on ROCm 7.2 when the graph is
// launched on an idle device (tests/test_e2e_gpu.py::test_hip_graph_...); kernel -> kernel edges are safe.
int zero_fill_async(void* ptr, size_t bytes, hipStream_t st);        // core.hip; bytes % 4 == 0
int zero_rows_async(float* ptr, int64_t rows, int64_t cols, int64_t ld, hipStream_t st);   // ptr[r*ld + c] = 0, c < cols

// elementwise.hip: out[col] = sum over parts of P[part * stride + col], fixed tree (deterministic)
int colsum_final_launch(int nparts, int F, const float* P, int64_t stride, float* out, hipStream_t st);

// softmax_adam.hip: softmax of the listed rows of P, IN PLACE (P holds their logits), + the first index of each row's maximum
int softmax_rows_indexed_launch(const int* rows_dev, int n_list, int C, float* P, int64_t ldp, int* amax, hipStream_t st);

// gemm.hip: C = act(sum_z W[z] + bias) [+ C], slabs added in index order
int splitk_reduce_launch(int64_t M, int64_t N, int nsplit, const float* W, int64_t ldw, float* C, int64_t ldc,
                         const float* bias, int act, int accumulate, hipStream_t st);

// gemm_bf16.hip
size_t gemm_bf16_tn_workspace_bytes(int64_t M, int64_t N, int64_t K);
size_t gemm_bf16_tn_dual_workspace_bytes(int64_t M, int64_t N0, int64_t N1, int64_t K);
int gemm_bf16_tn_dual_dispatch(int64_t M, int64_t N0, int64_t N1, int64_t K, const float* A, int64_t lda, const float* B0, int64_t ldb0,
                               const float* B1, int64_t ldb1, float* C0, int64_t ldc0, float* C1, int64_t ldc1, void* ws, size_t ws_bytes,
                               hipStream_t st);
int gemm_bf16_tn_dispatch(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B, int64_t ldb,
                          float* C, int64_t ldc, const float* bias, int act, int accumulate, void* ws, size_t ws_bytes,
                          hipStream_t st);
size_t gemm_bf16_workspace_bytes(int precision, int64_t N, int64_t K);
// `gate` (nullable): C += G * (1 - T) -- the highway block's carry gradient formed in the epilogue (fp32 C, no bias / activation /
// accumulate); shapes the whole-rows kernel does not take write it with geogcn_gate_carry_f32 first and accumulate onto it
size_t gemm_bf16_dual_workspace_bytes(int64_t N0, int64_t N1, int64_t K);
int gemm_bf16_dual_dispatch(int64_t M, int64_t N0, int64_t N1, int64_t K, const float* A, int64_t lda, const float* B0,
                            int64_t ldb0, const float* B1, int64_t ldb1, void* C0, int64_t ldc0, int c0_bf16, float* C1,
                            int64_t ldc1, const float* bias1, int act1, void* ws, size_t ws_bytes, hipStream_t st);
struct GateOps { const float* G; int64_t ldg; const float* T; int64_t ldt; };
// (round 6) C = A0 . op(B0) + A1 . op(B1) [+ C | + G (1 - T)] in one launch of the bf16 whole-rows kernel; 1 = shape not taken
size_t gemm_bf16_kcat_workspace_bytes(int64_t N, int64_t K0, int64_t K1);
bool gemm_bf16_kcat_native(int64_t N, int64_t K0, int64_t K1);
int gemm_bf16_kcat_dispatch(int transB, int64_t M, int64_t N, int64_t K0, int64_t K1, const float* A0, int64_t lda0, const float* B0,
                            int64_t ldb0, const float* A1, int64_t lda1, const float* B1, int64_t ldb1, float* C, int64_t ldc,
                            int accumulate, void* ws, size_t ws_bytes, hipStream_t st, const GateOps* gate);
int gemm_bf16_dispatch(int precision, int transB, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
                       const float* B, int64_t ldb, void* C, int64_t ldc, int c_bf16, const float* bias, int act,
                       int accumulate, void* ws, size_t ws_bytes, hipStream_t st, int panel_w = 0, int64_t panel_R = 0,
                       const GateOps* gate = nullptr);

}  // namespace geogcn
