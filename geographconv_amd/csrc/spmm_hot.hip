// CSR x dense SpMM with the HOT rows of the dense operand staged in LDS (gfx950) -- S.structured_dot(X, W0) for a
// bag-of-words X (reference gcnmodel.py:39-42).
//
// X's columns are Zipfian: at the TwitterUS shape the 120 most frequent vocabulary entries hold 55 % of the stored
// nonzeros, i.e. more than half of all gathered rows of W0 are the same 120 rows.  The plain gather kernel fetches
// each of them from L2 every time (21.5 M x 1.2 KB = 27 GB of L2 -> CU traffic: it runs at the L2 gather ceiling).
// Here one persistent 1024-thread workgroup per CU copies those rows into its 160 KB LDS once and serves every hot
// nonzero from there (ds_read_b128, conflict-free: 256 B per clock and CU); only the cold nonzeros go to L2.
//   * the caller reorders every CSR row [hot | cold] (each part in ascending column order) and stores, for a hot
//     entry, the LDS slot instead of the column (geogcn.h: geogcn_spmm_csr_hot_f32);
//   * a 16-lane group owns one row at a time, K4 float4 accumulators per lane, sequential fmaf in stored order
//     (hot part, then cold part): deterministic, bitwise reproducible;
//   * bias + tanh / sigmoid epilogue fused into the store.
#include "common.h"

#include <algorithm>

namespace geogcn {
namespace {

constexpr int kGroup = 16;
constexpr int kThreads = 1024;
constexpr int kGroups = kThreads / kGroup;
constexpr int kHotLdsBytes = 156 * 1024;

typedef float f32x4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float4 ld4g(const float* row, int q) {
    const f32x4v v = *(reinterpret_cast<const f32x4v*>(row) + q);
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void fma4(float4& acc, float a, const float4& b) {
    acc.x = fmaf(a, b.x, acc.x);
    acc.y = fmaf(a, b.y, acc.y);
    acc.z = fmaf(a, b.z, acc.z);
    acc.w = fmaf(a, b.w, acc.w);
}

struct HotArgs {
    int n_rows;
    const int* rowptr;
    const int* rowsplit;        // [rowptr[r], rowsplit[r]) hot (colidx = LDS slot), [rowsplit[r], rowptr[r+1]) cold
    const int* colidx;
    const float* val;
    const float* B; int64_t ldb;
    const int* hot_rows; int n_hot;
    float* C; int64_t ldc;
    int F;
    const float* bias;
};

template <int K4, int ACT>
__global__ __launch_bounds__(kThreads, 1) void spmm_hot_kernel(const HotArgs a) {
    extern __shared__ __attribute__((aligned(16))) float4 hot[];         // [n_hot][K4][16]
    const int lane = threadIdx.x % kGroup, g = threadIdx.x / kGroup;
    const int nF4 = (a.F + 3) >> 2;
    for (int i = threadIdx.x; i < a.n_hot * K4 * kGroup; i += kThreads) {
        const int slot = i / (K4 * kGroup), r = i % (K4 * kGroup);
        const int k = r / kGroup, l = r % kGroup;
        const int q = l + kGroup * k;
        hot[i] = (q < nF4) ? ld4g(a.B + (int64_t)a.hot_rows[slot] * a.ldb, q) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    const int n_tiles = (a.n_rows + kGroups - 1) / kGroups;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int row = tile * kGroups + g;
        if (row >= a.n_rows) continue;
        const int s = a.rowptr[row], h = a.rowsplit[row], e = a.rowptr[row + 1];
        float4 acc[K4];
#pragma unroll
        for (int k = 0; k < K4; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        // hot part: LDS
        for (int base = s; base < h; base += kGroup) {
            const int j = base + lane;
            int c = 0;
            float v = 0.f;
            if (j < h) {
                c = a.colidx[j];
                v = a.val[j];
            }
            const int cnt = min(kGroup, h - base);
            for (int t = 0; t < cnt; ++t) {
                const int slot = __shfl(c, t, kGroup);
                const float a0 = __shfl(v, t, kGroup);
                const float4* hr = hot + slot * (K4 * kGroup) + lane;
#pragma unroll
                for (int k = 0; k < K4; ++k) fma4(acc[k], a0, hr[k * kGroup]);
            }
        }
        // cold part: L2 / HBM gather, two nonzeros per trip
        for (int base = h; base < e; base += kGroup) {
            const int j = base + lane;
            int c = 0;
            float v = 0.f;
            if (j < e) {
                c = a.colidx[j];
                v = a.val[j];
            }
            const int cnt = min(kGroup, e - base);
            int t = 0;
            for (; t + 1 < cnt; t += 2) {
                const int c0 = __shfl(c, t, kGroup), c1 = __shfl(c, t + 1, kGroup);
                const float a0 = __shfl(v, t, kGroup), a1 = __shfl(v, t + 1, kGroup);
                const float* b0 = a.B + (int64_t)c0 * a.ldb;
                const float* b1 = a.B + (int64_t)c1 * a.ldb;
                float4 v0[K4], v1[K4];
#pragma unroll
                for (int k = 0; k < K4; ++k) {
                    const int q = lane + kGroup * k;
                    if (q < nF4) {
                        v0[k] = ld4g(b0, q);
                        v1[k] = ld4g(b1, q);
                    }
                }
#pragma unroll
                for (int k = 0; k < K4; ++k) {
                    const int q = lane + kGroup * k;
                    if (q < nF4) {
                        fma4(acc[k], a0, v0[k]);
                        fma4(acc[k], a1, v1[k]);
                    }
                }
            }
            if (t < cnt) {
                const int c0 = __shfl(c, t, kGroup);
                const float a0 = __shfl(v, t, kGroup);
                const float* b0 = a.B + (int64_t)c0 * a.ldb;
#pragma unroll
                for (int k = 0; k < K4; ++k) {
                    const int q = lane + kGroup * k;
                    if (q < nF4) fma4(acc[k], a0, ld4g(b0, q));
                }
            }
        }
        float4* out = reinterpret_cast<float4*>(a.C + (int64_t)row * a.ldc);
#pragma unroll
        for (int k = 0; k < K4; ++k) {
            const int q = lane + kGroup * k;
            if (q < nF4) {
                float o[4] = {acc[k].x, acc[k].y, acc[k].z, acc[k].w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int col = q * 4 + i;
                    if (col < a.F) {
                        float x = o[i];
                        if (a.bias) x += a.bias[col];
                        o[i] = apply_act<ACT>(x);
                    } else {
                        o[i] = 0.f;
                    }
                }
                out[q] = make_float4(o[0], o[1], o[2], o[3]);
            }
        }
    }
}

}  // namespace
}  // namespace geogcn

using namespace geogcn;

extern "C" {

int32_t geogcn_spmm_hot_capacity(int32_t F) {
    if (F <= 0 || F > 384) return 0;                    // (1024-thread workgroups: K4 <= 6 keeps the kernel <= 128 VGPRs)
    const int K4 = (int)cdiv(cdiv(F, 4), kGroup);
    return kHotLdsBytes / (K4 * kGroup * (int)sizeof(float4));
}

int geogcn_spmm_csr_hot_f32(int32_t n_rows, const int32_t* rowptr, const int32_t* rowsplit, const int32_t* colidx,
                            const float* val, const float* B, int64_t ldb, const int32_t* hot_rows, int32_t n_hot, float* C,
                            int64_t ldc, int32_t F, const float* bias, int32_t act, void* stream) {
    GEOGCN_REQUIRE(n_rows >= 0 && F > 0 && n_hot >= 0, GEOGCN_E_SIZE, "spmm_csr_hot_f32: bad sizes");
    if (n_rows == 0) return 0;
    GEOGCN_REQUIRE(rowptr && rowsplit && C && B && (n_hot == 0 || hot_rows), GEOGCN_E_NULL, "spmm_csr_hot_f32: null pointer");
    const int cap = geogcn_spmm_hot_capacity(F);
    GEOGCN_REQUIRE(cap > 0 && n_hot <= cap, GEOGCN_E_ARG, "spmm_csr_hot_f32: F=%d / n_hot=%d outside the LDS capacity (%d rows)", F,
                   n_hot, cap);
    const int F4 = (F + 3) / 4;
    GEOGCN_REQUIRE(ldb % 4 == 0 && ldc % 4 == 0 && ldb >= (int64_t)F4 * 4 && ldc >= (int64_t)F4 * 4 && aligned16(B) && aligned16(C),
                   GEOGCN_E_ALIGN, "spmm_csr_hot_f32: needs float4-addressable B and C");
    GEOGCN_REQUIRE(act >= GEOGCN_ACT_NONE && act <= GEOGCN_ACT_SIGMOID, GEOGCN_E_ARG, "spmm_csr_hot_f32: unknown act %d", act);
    const HotArgs a{n_rows, rowptr, rowsplit, colidx, val, B, ldb, hot_rows, n_hot, C, ldc, F, bias};
    const int K4 = (int)cdiv(F4, kGroup);
    const size_t lds = (size_t)std::max(1, n_hot) * K4 * kGroup * sizeof(float4);
    const int n_tiles = (int)cdiv(n_rows, kGroups);
    const dim3 grid((unsigned)std::min(n_tiles, kNumCU));
    hipStream_t st = (hipStream_t)stream;
#define GEOGCN_HOT(K, ACT)                                                                                          \
    do {                                                                                                            \
        auto kern = spmm_hot_kernel<K, ACT>;                                                                        \
        static bool attr_done = false;                                                                              \
        if (!attr_done) {                                                                                           \
            GEOGCN_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
            attr_done = true;                                                                                       \
        }                                                                                                           \
        hipLaunchKernelGGL(kern, grid, dim3(kThreads), lds, st, a);                                                 \
    } while (0)
#define GEOGCN_HOT_ACT(K)                                                  \
    case K:                                                                \
        if (act == GEOGCN_ACT_TANH) GEOGCN_HOT(K, GEOGCN_ACT_TANH);        \
        else if (act == GEOGCN_ACT_SIGMOID) GEOGCN_HOT(K, GEOGCN_ACT_SIGMOID); \
        else GEOGCN_HOT(K, GEOGCN_ACT_NONE);                               \
        break;
    switch (K4) {
        GEOGCN_HOT_ACT(1) GEOGCN_HOT_ACT(2) GEOGCN_HOT_ACT(3) GEOGCN_HOT_ACT(4) GEOGCN_HOT_ACT(5) GEOGCN_HOT_ACT(6)
        default:
            set_error("spmm_csr_hot_f32: F=%d not supported", F);
            return GEOGCN_E_ARG;
    }
#undef GEOGCN_HOT
#undef GEOGCN_HOT_ACT
    GEOGCN_LAUNCH_CHECK("spmm_hot_kernel");
    return 0;
}

}  // extern "C"
