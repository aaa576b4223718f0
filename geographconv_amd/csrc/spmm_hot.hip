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
//   * bias + tanh / sigmoid epilogue fused into the store;
//   * optionally (DROP) the dropout that follows this layer in the reference (lasagne DropoutLayer, gcnmodel.py:357)
//     rides in the same epilogue: the keep decisions come from the Philox stream of geogcn_dropout_mask_philox (same
//     counter = element index / 4 + offset: bit-identical masks) or from a caller-supplied mask, and the kernel stores
//     the activation H0 (the backward needs 1 - H0^2), the dropped copy Hd = H0 * keep / (1-p) and the mask byte --
//     instead of a separate mask kernel and an apply pass that re-reads H0 (528 MB at the TwitterUS shape).
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

// Buffer addressing for the cold gathers: one wave-uniform descriptor of B in SGPRs and ONE 32-bit byte offset per
// nonzero (column * pitch + lane * 16; the K4 pieces of a row differ by an immediate), instead of a 64-bit address per load.
// That is what lets a 1024-thread workgroup (128 VGPRs per lane) keep FOUR gathered rows in flight per trip instead of two
// -- the cold part of a row is a chain of dependent L2 round trips, and the trips halve.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t hot_rsrc(const float* base, int64_t bytes) {
    const uint32_t n = bytes <= 0 ? 0u : (bytes > 0x7FFFFFFFll ? 0x7FFFFFFFu : (uint32_t)bytes);
    const uint64_t b = reinterpret_cast<uint64_t>(base);
    const uint64_t bu = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(b >> 32)) << 32) |
                        (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)b);
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(bu), 0, __builtin_amdgcn_readfirstlane(n), 0x00020000);
}
__device__ __forceinline__ float4 ld4b(__amdgpu_buffer_rsrc_t r, uint32_t off) {
    const f32x4v v = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0));
    return make_float4(v.x, v.y, v.z, v.w);
}
constexpr uint32_t kHotOob = 0x80000000u;     // beyond any descriptor: the load returns zeros

struct HotArgs {
    int n_rows;
    int n_cols;                 // rows of B (buffer form: n_cols * ldb * 4 < 2^31)
    const int* rowptr;
    const int* rowsplit;        // [rowptr[r], rowsplit[r]) hot (colidx = LDS slot), [rowsplit[r], rowptr[r+1]) cold
    const int* colidx;
    const float* val;
    const float* B; int64_t ldb;
    const int* hot_rows; int n_hot;
    const int* row_order;       // nullable: position -> row (rows of similar length next to each other; see geogcn.h)
    float* C; int64_t ldc;
    int F;
    const float* bias;
    // DROP only
    float* Cd;                   // dropped copy, pitch ldc
    const uint8_t* mask_in;      // nullable: injected keep-mask [n_rows][F] (dense, no pitch)
    uint8_t* mask_out;           // nullable: keep-mask as generated / used
    float keep_prob, scale;
    uint64_t seed, offset;       // Philox stream position (quads), see geogcn_dropout_mask_philox
    const int64_t* calls;        // nullable: device-resident call counter (captured steps)
    int64_t per_call, base;
};

template <int K4, int ACT, int DROP = 0, int BUF = 1>
__global__ __launch_bounds__(kThreads, 1) void spmm_hot_kernel(const HotArgs a) {
    extern __shared__ __attribute__((aligned(16))) float4 hot[];         // [n_hot][K4][16]
    const int lane = threadIdx.x % kGroup;
    const int nF4 = (a.F + 3) >> 2;
    for (int i = threadIdx.x; i < a.n_hot * K4 * kGroup; i += kThreads) {
        const int slot = i / (K4 * kGroup), r = i % (K4 * kGroup);
        const int k = r / kGroup, l = r % kGroup;
        const int q = l + kGroup * k;
        hot[i] = (q < nF4) ? ld4g(a.B + (int64_t)a.hot_rows[slot] * a.ldb, q) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // Rows are handed out DYNAMICALLY inside the workgroup: workgroup b owns the contiguous rows [r_lo, r_hi) and every wave takes
    // the next four of them from a counter in LDS when it has finished its last four.  (A fixed stride gave every 16-lane group 27
    // rows of a log-normal length distribution -- the slowest of the 16,384 groups set the kernel's time.)  Which wave computes a
    // row never changes a value.
    __shared__ unsigned next_row;
    if (threadIdx.x == 0) next_row = 0;
    __syncthreads();
    const int r_lo = (int)((int64_t)a.n_rows * blockIdx.x / gridDim.x), r_hi = (int)((int64_t)a.n_rows * (blockIdx.x + 1) / gridDim.x);
    const int gw = (threadIdx.x & 63) / kGroup;          // group inside the wave
    while (true) {
        unsigned base = 0;
        if ((threadIdx.x & 63) == 0) base = atomicAdd(&next_row, 64 / kGroup);
        base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
        if (r_lo + (int)base >= r_hi) break;
        const int pos = r_lo + (int)base + gw;
        if (pos >= r_hi) continue;
        const int row = a.row_order ? a.row_order[pos] : pos;
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
        // cold part: L2 / HBM gather.  Buffer form: four nonzeros per trip (4 x K4 independent loads in flight per lane)
        if constexpr (BUF) {
            const __amdgpu_buffer_rsrc_t rs = hot_rsrc(a.B, (int64_t)a.n_cols * a.ldb * 4);
            const uint32_t ld4 = (uint32_t)a.ldb * 4u;
            constexpr int E = K4 <= 5 ? 4 : 3;          // (4 x 6 float4 in flight would spill at 128 VGPRs)
            for (int base = h; base < e; base += kGroup) {
                const int j = base + lane;
                uint32_t co = 0;
                float v = 0.f;
                if (j < e) {
                    co = (uint32_t)a.colidx[j] * ld4;          // byte offset of the gathered row
                    v = a.val[j];
                }
                const int cnt = min(kGroup, e - base);
                // the last piece of a row may lie beyond F (lanes whose float4 index is >= nF4): those loads are sent out of range
                const uint32_t lane_off = (uint32_t)lane * 16u;
                const bool last_ok = lane + kGroup * (K4 - 1) < nF4;
                int t = 0;
                for (; t + E <= cnt; t += E) {
                    uint32_t off[E];
                    float av[E];
#pragma unroll
                    for (int q = 0; q < E; ++q) {
                        off[q] = (uint32_t)__shfl((int)co, t + q, kGroup) + lane_off;
                        av[q] = __shfl(v, t + q, kGroup);
                    }
                    float4 vv[E][K4];
#pragma unroll
                    for (int q = 0; q < E; ++q)
#pragma unroll
                        for (int k = 0; k < K4; ++k)
                            vv[q][k] = ld4b(rs, (k == K4 - 1 && !last_ok) ? kHotOob : off[q] + (uint32_t)(k * kGroup * 16));
#pragma unroll
                    for (int q = 0; q < E; ++q)            // stored order: nonzero t, t+1, ... into every accumulator
#pragma unroll
                        for (int k = 0; k < K4; ++k) fma4(acc[k], av[q], vv[q][k]);
                }
                for (; t < cnt; ++t) {
                    const uint32_t o0 = (uint32_t)__shfl((int)co, t, kGroup) + lane_off;
                    const float a0 = __shfl(v, t, kGroup);
                    float4 v0[K4];
#pragma unroll
                    for (int k = 0; k < K4; ++k) v0[k] = ld4b(rs, (k == K4 - 1 && !last_ok) ? kHotOob : o0 + (uint32_t)(k * kGroup * 16));
#pragma unroll
                    for (int k = 0; k < K4; ++k) fma4(acc[k], a0, v0[k]);
                }
            }
        } else
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
        uint64_t drop_offset = a.offset;
        if constexpr (DROP) {
            if (a.calls) drop_offset = (uint64_t)((*a.calls * a.per_call + a.base) / 4);
        }
#pragma unroll
        for (int k = 0; k < K4; ++k) {
            const int q = lane + kGroup * k;
            if (q < nF4) {
                float o[4] = {acc[k].x, acc[k].y, acc[k].z, acc[k].w};
                const float4 b4 = load_bias4(a.bias, q * 4, a.F, (reinterpret_cast<uintptr_t>(a.bias) & 15u) == 0);
                const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int col = q * 4 + i;
                    if (col < a.F) {
                        float x = o[i];
                        if (a.bias) x += bb[i];
                        o[i] = apply_act<ACT>(x);
                    } else {
                        o[i] = 0.f;
                    }
                }
                out[q] = make_float4(o[0], o[1], o[2], o[3]);
                if constexpr (DROP) {
                    // F % 4 == 0 (checked by the entry point): the four elements are one Philox counter and one aligned
                    // word of the byte mask
                    const int64_t e0 = (int64_t)row * a.F + q * 4;
                    uint32_t m4;
                    if (a.mask_in) {
                        m4 = *reinterpret_cast<const uint32_t*>(a.mask_in + e0);
                    } else {
                        float u[4];
                        philox_uniform4(a.seed, (uint64_t)(e0 >> 2) + drop_offset, u);
                        m4 = (u[0] < a.keep_prob ? 1u : 0u) | (u[1] < a.keep_prob ? 0x100u : 0u) |
                             (u[2] < a.keep_prob ? 0x10000u : 0u) | (u[3] < a.keep_prob ? 0x1000000u : 0u);
                    }
                    if (a.mask_out) *reinterpret_cast<uint32_t*>(a.mask_out + e0) = m4;
                    // (same arithmetic as dropout_apply_kernel: x * ((float)mask * scale))
                    reinterpret_cast<float4*>(a.Cd + (int64_t)row * a.ldc)[q] =
                        make_float4(o[0] * ((float)(m4 & 0xffu) * a.scale), o[1] * ((float)((m4 >> 8) & 0xffu) * a.scale),
                                    o[2] * ((float)((m4 >> 16) & 0xffu) * a.scale), o[3] * ((float)(m4 >> 24) * a.scale));
                }
            }
        }
    }
}

// which form runs: the buffer form whenever B is addressable with 31 bits -- except the widest kernel with the dropout epilogue,
// whose epilogue already fills the 128 registers of a 1024-thread workgroup (the buffer form would spill there)
template <int K4, int ACT, int DROP>
auto hot_kernel_for(bool buf) -> void (*)(const HotArgs) {
    if constexpr (K4 == 6 && DROP) return spmm_hot_kernel<K4, ACT, DROP, 0>;
    else return buf ? spmm_hot_kernel<K4, ACT, DROP, 1> : spmm_hot_kernel<K4, ACT, DROP, 0>;
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

static int spmm_hot_launch(const char* fn, HotArgs a, int32_t n_hot, int32_t act, int drop, hipStream_t st) {
    const int n_rows = a.n_rows, F = a.F;
    GEOGCN_REQUIRE(n_rows >= 0 && F > 0 && n_hot >= 0 && a.n_cols >= 0, GEOGCN_E_SIZE, "%s: bad sizes", fn);
    if (n_rows == 0) return 0;
    GEOGCN_REQUIRE(a.rowptr && a.rowsplit && a.C && a.B && (n_hot == 0 || a.hot_rows), GEOGCN_E_NULL, "%s: null pointer", fn);
    const int cap = geogcn_spmm_hot_capacity(F);
    GEOGCN_REQUIRE(cap > 0 && n_hot <= cap, GEOGCN_E_ARG, "%s: F=%d / n_hot=%d outside the LDS capacity (%d rows)", fn, F, n_hot, cap);
    const int F4 = (F + 3) / 4;
    GEOGCN_REQUIRE(a.ldb % 4 == 0 && a.ldc % 4 == 0 && a.ldb >= (int64_t)F4 * 4 && a.ldc >= (int64_t)F4 * 4 && aligned16(a.B) &&
                       aligned16(a.C), GEOGCN_E_ALIGN, "%s: needs float4-addressable B and C", fn);
    GEOGCN_REQUIRE(act >= GEOGCN_ACT_NONE && act <= GEOGCN_ACT_SIGMOID, GEOGCN_E_ARG, "%s: unknown act %d", fn, act);
    const int K4 = (int)cdiv(F4, kGroup);
    const size_t lds = (size_t)std::max(1, n_hot) * K4 * kGroup * sizeof(float4);
    const int n_tiles = (int)cdiv(n_rows, kGroups);
    const dim3 grid((unsigned)std::min(n_tiles, kNumCU));
    // buffer form when every byte of B is within a 31-bit offset (always at the reference's shapes: a vocabulary x hidden weight)
    const bool buf = a.n_cols > 0 && (int64_t)a.n_cols * a.ldb * 4 < 0x7FFFFFFFll;
#define GEOGCN_HOT(K, ACT, DROP)                                                                                    \
    do {                                                                                                            \
        auto kern = hot_kernel_for<K, ACT, DROP>(buf);                                                              \
        static LdsAttrOnce lds_once[2];                                                                  \
        if (const int rc_ = lds_once[buf].ensure((const void*)kern, (int)(kHotLdsBytes))) return rc_; \
        hipLaunchKernelGGL(kern, grid, dim3(kThreads), lds, st, a);                                                 \
    } while (0)
#define GEOGCN_HOT_ACT(K)                                                     \
    case K:                                                                   \
        if (drop) {           /* the layer the dropout follows is tanh (gcnmodel.py:347,357); linear for completeness */ \
            if (act == GEOGCN_ACT_TANH) GEOGCN_HOT(K, GEOGCN_ACT_TANH, 1);    \
            else GEOGCN_HOT(K, GEOGCN_ACT_NONE, 1);                           \
        } else if (act == GEOGCN_ACT_TANH) GEOGCN_HOT(K, GEOGCN_ACT_TANH, 0); \
        else if (act == GEOGCN_ACT_SIGMOID) GEOGCN_HOT(K, GEOGCN_ACT_SIGMOID, 0); \
        else GEOGCN_HOT(K, GEOGCN_ACT_NONE, 0);                               \
        break;
    switch (K4) {
        GEOGCN_HOT_ACT(1) GEOGCN_HOT_ACT(2) GEOGCN_HOT_ACT(3) GEOGCN_HOT_ACT(4) GEOGCN_HOT_ACT(5) GEOGCN_HOT_ACT(6)
        default:
            set_error("%s: F=%d not supported", fn, F);
            return GEOGCN_E_ARG;
    }
#undef GEOGCN_HOT
#undef GEOGCN_HOT_ACT
    GEOGCN_LAUNCH_CHECK("spmm_hot_kernel");
    return 0;
}

int geogcn_spmm_csr_hot_f32(int32_t n_rows, int32_t n_cols, const int32_t* rowptr, const int32_t* rowsplit, const int32_t* colidx,
                            const float* val, const float* B, int64_t ldb, const int32_t* hot_rows, int32_t n_hot,
                            const int32_t* row_order, float* C, int64_t ldc, int32_t F, const float* bias, int32_t act, void* stream) {
    HotArgs a{};
    a.n_rows = n_rows; a.n_cols = n_cols; a.row_order = row_order; a.rowptr = rowptr; a.rowsplit = rowsplit; a.colidx = colidx; a.val = val; a.B = B; a.ldb = ldb;
    a.hot_rows = hot_rows; a.n_hot = n_hot; a.C = C; a.ldc = ldc; a.F = F; a.bias = bias;
    return spmm_hot_launch("spmm_csr_hot_f32", a, n_hot, act, 0, (hipStream_t)stream);
}

int geogcn_spmm_csr_hot_dropout_f32(int32_t n_rows, int32_t n_cols, const int32_t* rowptr, const int32_t* rowsplit, const int32_t* colidx,
                                    const float* val, const float* B, int64_t ldb, const int32_t* hot_rows, int32_t n_hot,
                                    const int32_t* row_order, float* C, float* Cd, int64_t ldc, int32_t F, const float* bias, int32_t act,
                                    float p_drop,
                                    const uint8_t* mask_in, uint8_t* mask_out, uint64_t seed, uint64_t offset,
                                    const int64_t* calls_dev, int64_t per_call_elems, int64_t base_elems, void* stream) {
    const char* fn = "spmm_csr_hot_dropout_f32";
    GEOGCN_REQUIRE(p_drop >= 0.f && p_drop < 1.f, GEOGCN_E_ARG, "%s: p=%f outside [0,1)", fn, p_drop);
    GEOGCN_REQUIRE(act == GEOGCN_ACT_TANH || act == GEOGCN_ACT_NONE, GEOGCN_E_ARG, "%s: act must be tanh or none (got %d)", fn, act);
    GEOGCN_REQUIRE(per_call_elems >= 0 && base_elems >= 0, GEOGCN_E_SIZE, "%s: negative stream position", fn);
    if (n_rows == 0) return 0;
    GEOGCN_REQUIRE(Cd && (mask_in || mask_out), GEOGCN_E_NULL, "%s: null pointer (Cd, and one of mask_in / mask_out)", fn);
    GEOGCN_REQUIRE(F > 0 && F % 4 == 0 && aligned16(Cd) && (reinterpret_cast<uintptr_t>(mask_in) & 3u) == 0 &&
                       (reinterpret_cast<uintptr_t>(mask_out) & 3u) == 0,
                   GEOGCN_E_ALIGN, "%s: needs F %% 4 == 0 (one Philox counter / one mask word per float4), a 16-byte aligned Cd "
                   "and 4-byte aligned masks (F=%d)", fn, F);
    HotArgs a{};
    a.n_rows = n_rows; a.n_cols = n_cols; a.rowptr = rowptr; a.rowsplit = rowsplit; a.colidx = colidx; a.val = val; a.B = B; a.ldb = ldb;
    a.hot_rows = hot_rows; a.n_hot = n_hot; a.row_order = row_order; a.C = C; a.ldc = ldc; a.F = F; a.bias = bias;
    a.Cd = Cd; a.mask_in = mask_in; a.mask_out = mask_out; a.keep_prob = 1.0f - p_drop; a.scale = 1.0f / (1.0f - p_drop);
    a.seed = seed; a.offset = offset; a.calls = calls_dev; a.per_call = per_call_elems; a.base = base_elems;
    return spmm_hot_launch(fn, a, n_hot, act, 1, (hipStream_t)stream);
}

}  // extern "C"
