// Streaming elementwise / reduction kernels of the GCN hot path (gfx950): the work Theano runs
// as fused Elemwise / Sum{axis=0} / AdvancedSubtensor1 / MRG binomial C thunks.
//   highway mix + gradients        reference gcnmodel.py:252-266, :268-288
//   bias + nonlinearity            reference gcnmodel.py:41-42, :132-136, :155-157
//   dropout                        reference gcnmodel.py:357 (lasagne DropoutLayer)
// All are HBM-bound: 16-byte accesses, grid-stride over (row, float4) with the pad columns
// [F, ld) written as zero, no LDS staging (no reuse to capture).
#include "common.h"

#include <stdlib.h>

#include <algorithm>

namespace geogcn {
namespace {

constexpr int TPB = 256;

inline unsigned stream_grid(int64_t work_items) {
    // ~8 blocks of 256 threads per CU is enough to saturate HBM; grid-stride the rest
    const int64_t blocks = cdiv(work_items, TPB);
    return (unsigned)std::max<int64_t>(1, std::min<int64_t>(blocks, (int64_t)kNumCU * 8));
}

struct Idx2 {
    int64_t row;
    int q;       // float4 index within the row
};

__device__ __forceinline__ float4 mask_pad(float4 v, int col0, int F) {
    if (col0 + 3 < F) return v;
    if (col0 + 0 >= F) v.x = 0.f;
    if (col0 + 1 >= F) v.y = 0.f;
    if (col0 + 2 >= F) v.z = 0.f;
    v.w = 0.f;
    return v;
}

template <int ACT>
__global__ __launch_bounds__(TPB) void bias_act_kernel(int64_t n, int F, int F4, const float* __restrict__ X,
                                                       int64_t ldx, const float* __restrict__ bias,
                                                       float* __restrict__ Y, int64_t ldy) {
    const int64_t total = n * F4;
    for (int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x; e < total; e += (int64_t)gridDim.x * TPB) {
        const int64_t row = e / F4;
        const int q = (int)(e - row * F4);
        const int c0 = q * 4;
        float4 v = *reinterpret_cast<const float4*>(X + row * ldx + c0);
        float o[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (c0 + i < F) {
                float x = o[i];
                if (bias) x += bias[c0 + i];
                o[i] = apply_act<ACT>(x);
            } else {
                o[i] = 0.f;
            }
        }
        *reinterpret_cast<float4*>(Y + row * ldy + c0) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// fp32 -> bfloat16, round to nearest even (NaN stays NaN); whole pitch of Y written, pads as zeros
__device__ __forceinline__ unsigned bf16_rne(float x) {
    const unsigned u = __float_as_uint(x);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;      // quiet NaN
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
__global__ __launch_bounds__(TPB) void cast_bf16_kernel(int64_t n, int F, int Y4, const float* __restrict__ X, int64_t ldx,
                                                        uint16_t* __restrict__ Y, int64_t ldy) {
    const int64_t total = n * Y4;
    for (int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x; e < total; e += (int64_t)gridDim.x * TPB) {
        const int64_t row = e / Y4;
        const int c0 = (int)(e - row * Y4) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c0 < F) v = *reinterpret_cast<const float4*>(X + row * ldx + c0);      // X pads are zero by convention
        uint2 o;
        o.x = bf16_rne(v.x) | (bf16_rne(v.y) << 16);
        o.y = bf16_rne(v.z) | (bf16_rne(v.w) << 16);
        *reinterpret_cast<uint2*>(Y + row * ldy + c0) = o;
    }
}

// Hout = T*Hc + (1-T)*H   (gcnmodel.py:266, same association as the reference expression)
__global__ __launch_bounds__(TPB) void highway_fwd_kernel(int64_t total4, const float4* __restrict__ T,
                                                          const float4* __restrict__ Hc,
                                                          const float4* __restrict__ H, float4* __restrict__ Out) {
    for (int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x; e < total4; e += (int64_t)gridDim.x * TPB) {
        const float4 t = T[e], hc = Hc[e], h = H[e];
        float4 o;
        o.x = t.x * hc.x + (1.0f - t.x) * h.x;
        o.y = t.y * hc.y + (1.0f - t.y) * h.y;
        o.z = t.z * hc.z + (1.0f - t.z) * h.z;
        o.w = t.w * hc.w + (1.0f - t.w) * h.w;
        Out[e] = o;
    }
}

// the three gradients of the gating mix, every operation rounded on its own (no fused multiply-add): the four kernels below
// -- fp32 / bf16 dS, with / without column sums -- must produce the SAME fp32 values (a contraction chosen differently in one
// instantiation moved dS by an ulp and its bf16 rounding across a tie), and this is the arithmetic of the NumPy restatement
__device__ __forceinline__ void hw_grad(float g, float t, float hc, float h, float& s, float& u, float& c) {
#pragma clang fp contract(off)
    s = (g * t) * (1.0f - hc * hc);
    u = ((g * (hc - h)) * t) * (1.0f - t);
    c = g * (1.0f - t);
}

// S16: dS is stored as bfloat16 (round to nearest even, the bits geogcn_cast_bf16_f32 would produce), pitch ld4_dS in units of
// four elements, the WHOLE pitch written (pads as zeros) -- the bf16 configuration's A^T . dS gathers it as it is, so the
// fp32 dS and the separate cast pass disappear (the bias gradient below is still the column sum of the fp32 values)
// (gfx950's v_cvt_pk_bf16_f32: two values per instruction, the bits of bf16_rne for every finite input -- the integer sequence
//  costs ~10 VALU instructions per value, which this latency-bound kernel feels: 1.57 against 1.29 ms at 600 wide)
typedef float hw_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 hw_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint2 pack_bf16x4(const float4& s) {
    const hw_f32x2 lo = {s.x, s.y}, hi = {s.z, s.w};
    uint2 o;
    o.x = __builtin_bit_cast(uint32_t, __builtin_convertvector(lo, hw_bf16x2));
    o.y = __builtin_bit_cast(uint32_t, __builtin_convertvector(hi, hw_bf16x2));
    return o;
}

template <bool S16>
__global__ __launch_bounds__(TPB) void highway_bwd_kernel(int64_t n, int ld4, const float4* __restrict__ G,
                                                          const float4* __restrict__ T, const float4* __restrict__ Hc,
                                                          const float4* __restrict__ H, void* __restrict__ dSv,
                                                          int ld4_dS, float4* __restrict__ dU, float4* __restrict__ dHc) {
    float4* dS = (float4*)dSv;
    const int64_t total4 = n * ld4;
    for (int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x; e < total4; e += (int64_t)gridDim.x * TPB) {
        const float4 g = G[e], t = T[e], hc = Hc[e], h = H[e];
        float4 s, u, c;
#define GEOGCN_HW(m) hw_grad(g.m, t.m, hc.m, h.m, s.m, u.m, c.m);
        GEOGCN_HW(x) GEOGCN_HW(y) GEOGCN_HW(z) GEOGCN_HW(w)
#undef GEOGCN_HW
        int64_t es = e;
        if (S16 || ld4_dS != ld4) {
            const int64_t row = e / ld4;
            const int q = (int)(e - row * ld4);
            es = row * ld4_dS + q;
            if constexpr (S16) {
                for (int pq = ld4 + q; pq < ld4_dS; pq += ld4) ((uint2*)dSv)[row * ld4_dS + pq] = make_uint2(0u, 0u);
            }
        }
        if constexpr (S16) ((uint2*)dSv)[es] = pack_bf16x4(s);
        else dS[es] = s;
        dU[e] = u;
        if (dHc) dHc[e] = c;          // (NULL: the carry is formed where it is consumed -- geogcn_gemm_kcat_gated_f32)
    }
}

// Same arithmetic, plus the column sums of dS and dU (the two bias gradients) in the same pass: block b owns
// a contiguous chunk of rows, thread (ri, q) walks float4 column q of every rpi-th row; the per-thread sums are
// combined over ri in fixed order and written as one partial row per block (summed by colsum_final_kernel).
template <bool S16>
__global__ __launch_bounds__(TPB) void highway_bwd_colsum_kernel(int64_t n, int ld4, const float4* __restrict__ G,
                                                                 const float4* __restrict__ T, const float4* __restrict__ Hc,
                                                                 const float4* __restrict__ H, void* __restrict__ dSv,
                                                                 int ld4_dS, float4* __restrict__ dU, float4* __restrict__ dHc,
                                                                 int64_t rows_per_block, float4* __restrict__ P) {
    float4* dS = (float4*)dSv;
    __shared__ float4 red[2][TPB];
    const int W = ld4, rpi = TPB / W;
    const int q = threadIdx.x % W, ri = threadIdx.x / W;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = min(n, r0 + rows_per_block);
    float4 aS = make_float4(0.f, 0.f, 0.f, 0.f), aU = aS;
    if (ri < rpi) {
        // two rows per trip: eight loads in flight per thread instead of four (the sums are still taken in row order)
        auto one = [&](int64_t row, const float4& g, const float4& t, const float4& hc, const float4& h) {
            const int64_t e = row * ld4 + q;
            float4 s, u, c;
#define GEOGCN_HW(m)                              \
    hw_grad(g.m, t.m, hc.m, h.m, s.m, u.m, c.m);  \
    aS.m += s.m;                                  \
    aU.m += u.m;
            GEOGCN_HW(x) GEOGCN_HW(y) GEOGCN_HW(z) GEOGCN_HW(w)
#undef GEOGCN_HW
            if constexpr (S16) {
                ((uint2*)dSv)[row * ld4_dS + q] = pack_bf16x4(s);
                for (int pq = W + q; pq < ld4_dS; pq += W) ((uint2*)dSv)[row * ld4_dS + pq] = make_uint2(0u, 0u);
            } else {
                dS[row * ld4_dS + q] = s;
            }
            dU[e] = u;
            if (dHc) dHc[e] = c;
        };
        int64_t row = r0 + ri;
        for (; row + rpi < r1; row += 2 * rpi) {
            const int64_t e0 = row * ld4 + q, e1 = (row + rpi) * ld4 + q;
            const float4 g0 = G[e0], t0 = T[e0], hc0 = Hc[e0], h0 = H[e0];
            const float4 g1 = G[e1], t1 = T[e1], hc1 = Hc[e1], h1 = H[e1];
            one(row, g0, t0, hc0, h0);
            one(row + rpi, g1, t1, hc1, h1);
        }
        if (row < r1) {
            const int64_t e = row * ld4 + q;
            one(row, G[e], T[e], Hc[e], H[e]);
        }
    }
    red[0][threadIdx.x] = aS;
    red[1][threadIdx.x] = aU;
    __syncthreads();
    if (ri == 0) {
        for (int k = 1; k < rpi; ++k) {
            const float4 a = red[0][k * W + q], b = red[1][k * W + q];
            aS.x += a.x; aS.y += a.y; aS.z += a.z; aS.w += a.w;
            aU.x += b.x; aU.y += b.y; aU.z += b.z; aU.w += b.w;
        }
        P[(int64_t)blockIdx.x * 2 * W + q] = aS;
        P[(int64_t)blockIdx.x * 2 * W + W + q] = aU;
    }
}

// the carry gradient of the gating mix on its own: out = G * (1 - T), the arithmetic of hw_grad's `c`
__global__ __launch_bounds__(TPB) void gate_carry_kernel(int64_t n, int F4, const float* __restrict__ G, int64_t ldg,
                                                         const float* __restrict__ T, int64_t ldt, float* __restrict__ out, int64_t ldo) {
#pragma clang fp contract(off)
    const int64_t total = n * F4;
    for (int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x; e < total; e += (int64_t)gridDim.x * TPB) {
        const int64_t row = e / F4;
        const int c0 = (int)(e - row * F4) * 4;
        const float4 g = *reinterpret_cast<const float4*>(G + row * ldg + c0), t = *reinterpret_cast<const float4*>(T + row * ldt + c0);
        *reinterpret_cast<float4*>(out + row * ldo + c0) =
            make_float4(g.x * (1.0f - t.x), g.y * (1.0f - t.y), g.z * (1.0f - t.z), g.w * (1.0f - t.w));
    }
}

template <int ACT>
__global__ __launch_bounds__(TPB) void act_bwd_kernel(int64_t n, int F, int F4, const float* __restrict__ G,
                                                       const float* __restrict__ Y, int64_t ld,
                                                       const uint8_t* __restrict__ mask, float scale,
                                                       float* __restrict__ dS, int64_t ld_dS) {
    const int64_t total = n * F4;
    for (int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x; e < total; e += (int64_t)gridDim.x * TPB) {
        const int64_t row = e / F4;
        const int q = (int)(e - row * F4);
        const int c0 = q * 4;
        const float4 g = *reinterpret_cast<const float4*>(G + row * ld + c0);
        const float4 y = *reinterpret_cast<const float4*>(Y + row * ld + c0);
        float go[4] = {g.x, g.y, g.z, g.w};
        const float yo[4] = {y.x, y.y, y.z, y.w};
        float o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float gg = go[i];
            if (mask) gg = (c0 + i < F) ? gg * ((float)mask[row * F + c0 + i] * scale) : 0.f;
            if constexpr (ACT == GEOGCN_ACT_TANH) o[i] = tanh_bwd_val(gg, yo[i]);
            else if constexpr (ACT == GEOGCN_ACT_SIGMOID) o[i] = gg * (yo[i] * (1.0f - yo[i]));
            else if constexpr (ACT == GEOGCN_ACT_SELU)      // y > 0: scale; else scale*alpha*exp(x) = y + scale*alpha
                o[i] = gg * (yo[i] > 0.f ? 1.0507009873554805f : yo[i] + 1.0507009873554805f * 1.6732632423543772f);
            else if constexpr (ACT == GEOGCN_ACT_RELU) o[i] = yo[i] > 0.f ? gg : 0.f;
            else o[i] = gg;
        }
        *reinterpret_cast<float4*>(dS + row * ld_dS + c0) = mask_pad(make_float4(o[0], o[1], o[2], o[3]), c0, F);
    }
}

// act_bwd with the bias gradient (column sums of dS) in the same pass -- same block / thread layout and the same
// fixed summation order as highway_bwd_colsum_kernel
// Column sums alone, in the order the fused activation-gradient kernels (act_bwd_colsum_kernel / highway_bwd_colsum_kernel) take
// them -- block b owns a contiguous chunk of rows, thread (ri, q) adds float4 column q of every rpi-th row, the per-thread sums are
// combined over ri in order, one partial row per block -- so that a bias gradient formed from a matrix some other kernel produced
// (the product's epilogue of round 4) has the bits the fused pass would have given it
__global__ __launch_bounds__(TPB) void colsum_rowblocks_kernel(int64_t n, int F, int W, const float4* __restrict__ X, int ldx4,
                                                               int64_t rows_per_block, float4* __restrict__ P) {
    __shared__ float4 red[TPB];
    const int rpi = TPB / W;
    const int q = threadIdx.x % W, ri = threadIdx.x / W;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = min(n, r0 + rows_per_block);
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ri < rpi) {
        // four rows per trip: four loads in flight per thread, added in row order (the order of the fused kernels)
        int64_t row = r0 + ri;
        for (; row + 3 * (int64_t)rpi < r1; row += 4 * (int64_t)rpi) {
            const float4 x0 = X[row * ldx4 + q], x1 = X[(row + rpi) * ldx4 + q];
            const float4 x2 = X[(row + 2 * (int64_t)rpi) * ldx4 + q], x3 = X[(row + 3 * (int64_t)rpi) * ldx4 + q];
            const float4 s0 = mask_pad(x0, q * 4, F), s1 = mask_pad(x1, q * 4, F), s2 = mask_pad(x2, q * 4, F), s3 = mask_pad(x3, q * 4, F);
            a.x += s0.x; a.y += s0.y; a.z += s0.z; a.w += s0.w;
            a.x += s1.x; a.y += s1.y; a.z += s1.z; a.w += s1.w;
            a.x += s2.x; a.y += s2.y; a.z += s2.z; a.w += s2.w;
            a.x += s3.x; a.y += s3.y; a.z += s3.z; a.w += s3.w;
        }
        for (; row < r1; row += rpi) {
            const float4 s = mask_pad(X[row * ldx4 + q], q * 4, F);
            a.x += s.x; a.y += s.y; a.z += s.z; a.w += s.w;
        }
    }
    red[threadIdx.x] = a;
    __syncthreads();
    if (ri == 0) {
        for (int k = 1; k < rpi; ++k) {
            const float4 b = red[k * W + q];
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        P[(int64_t)blockIdx.x * W + q] = a;
    }
}

template <int ACT>
__global__ __launch_bounds__(TPB) void act_bwd_colsum_kernel(int64_t n, int F, int ld4, const float4* __restrict__ G,
                                                              const float4* __restrict__ Y,
                                                              const uint8_t* __restrict__ mask, float scale,
                                                              float4* __restrict__ dS, int ld4_dS, int64_t rows_per_block,
                                                              float4* __restrict__ P) {
    __shared__ float4 red[TPB];
    const int W = ld4, rpi = TPB / W;
    const int q = threadIdx.x % W, ri = threadIdx.x / W;
    const int c0 = q * 4;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = min(n, r0 + rows_per_block);
    float4 aS = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ri < rpi) {
        for (int64_t row = r0 + ri; row < r1; row += rpi) {
            const float4 g = G[row * ld4 + q], y = Y[row * ld4 + q];
            float go[4] = {g.x, g.y, g.z, g.w};
            const float yo[4] = {y.x, y.y, y.z, y.w};
            float o[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float gg = go[i];
                if (mask) gg = (c0 + i < F) ? gg * ((float)mask[row * F + c0 + i] * scale) : 0.f;
                if constexpr (ACT == GEOGCN_ACT_TANH) o[i] = tanh_bwd_val(gg, yo[i]);
                else if constexpr (ACT == GEOGCN_ACT_SIGMOID) o[i] = gg * (yo[i] * (1.0f - yo[i]));
                else o[i] = gg;
            }
            const float4 s = mask_pad(make_float4(o[0], o[1], o[2], o[3]), c0, F);
            dS[row * ld4_dS + q] = s;
            aS.x += s.x; aS.y += s.y; aS.z += s.z; aS.w += s.w;
        }
    }
    red[threadIdx.x] = aS;
    __syncthreads();
    if (ri == 0) {
        for (int k = 1; k < rpi; ++k) {
            const float4 a = red[k * W + q];
            aS.x += a.x; aS.y += a.y; aS.z += a.z; aS.w += a.w;
        }
        P[(int64_t)blockIdx.x * W + q] = aS;
    }
}

__global__ __launch_bounds__(TPB) void dropout_apply_kernel(int64_t n, int F, int F4, const float* __restrict__ X,
                                                            int64_t ld, const uint8_t* __restrict__ mask,
                                                            float scale, float* __restrict__ Y) {
    const int64_t total = n * F4;
    for (int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x; e < total; e += (int64_t)gridDim.x * TPB) {
        const int64_t row = e / F4;
        const int q = (int)(e - row * F4);
        const int c0 = q * 4;
        const float4 x = *reinterpret_cast<const float4*>(X + row * ld + c0);
        const float xo[4] = {x.x, x.y, x.z, x.w};
        float o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
            o[i] = (c0 + i < F) ? xo[i] * ((float)mask[row * F + c0 + i] * scale) : 0.f;
        *reinterpret_cast<float4*>(Y + row * ld + c0) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// ---- dropout masks: Philox4x32-10 (common.h), counter = element index / 4, key = seed ------------------
// `calls` (nullable): device-resident call counter of a captured step (hipGraph replays cannot change kernel
// arguments); the stream position is then (calls * per_call + base) / 4 quads, as the host computes it.
__global__ __launch_bounds__(TPB) void dropout_mask_kernel(int64_t total, float keep_prob, uint64_t seed,
                                                           uint64_t offset, const int64_t* __restrict__ calls,
                                                           int64_t per_call, int64_t base, uint8_t* __restrict__ mask) {
    if (calls) offset = (uint64_t)((*calls * per_call + base) / 4);
    const int64_t quads = (total + 3) / 4;
    for (int64_t qd = (int64_t)blockIdx.x * TPB + threadIdx.x; qd < quads; qd += (int64_t)gridDim.x * TPB) {
        float u[4];
        philox_uniform4(seed, (uint64_t)qd + offset, u);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t e = qd * 4 + i;
            if (e < total) mask[e] = (u[i] < keep_prob) ? 1 : 0;
        }
    }
}

// ---- dropout on the VALUES of a sparse matrix (reference gcnmodel.py:44-70, SparseInputDropoutLayer) -------------
// Element (i, j) of the logical matrix is kept iff the Philox draw keyed by its POSITION e = i * n_cols + j is below
// the keep probability -- the same bit whichever device layout holds the element (CSR of X, CSR of the tail of X^T,
// dense head panel), so the three layouts stay one matrix.
__device__ __forceinline__ float philox_uniform_at(uint64_t seed, uint64_t call_quads, int64_t e) {
    const uint64_t ctr = (uint64_t)(e >> 2) + call_quads;
    uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u};
    uint32_t k[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
#pragma unroll
    for (int r = 0; r < 10; ++r) philox_round(c, k);
    return (float)(c[e & 3] >> 8) * (1.0f / 16777216.0f);
}
__global__ __launch_bounds__(TPB) void dropout_csr_kernel(int64_t n_rows, const int* __restrict__ rowptr,
                                                          const int* __restrict__ colidx, const float* __restrict__ vin,
                                                          float* __restrict__ vout, int64_t n_cols_logical, int transposed,
                                                          float keep, float scale, uint64_t seed, uint64_t call_quads) {
    const int64_t row = (int64_t)blockIdx.x * (TPB / kWave) + threadIdx.x / kWave;
    if (row >= n_rows) return;
    const int lane = threadIdx.x % kWave;
    for (int j = rowptr[row] + lane; j < rowptr[row + 1]; j += kWave) {
        const int64_t col = colidx[j];
        const int64_t e = transposed ? col * n_cols_logical + row : row * n_cols_logical + col;
        vout[j] = philox_uniform_at(seed, call_quads, e) < keep ? vin[j] * scale : 0.f;
    }
}
__global__ __launch_bounds__(TPB) void dropout_panel_kernel(int64_t n, int K, const float* __restrict__ Pin, int64_t ld,
                                                            const int* __restrict__ head_idx, int64_t n_cols_logical,
                                                            float keep, float scale, uint64_t seed, uint64_t call_quads,
                                                            float* __restrict__ Pout) {
    const int64_t total = n * K;
    for (int64_t t = (int64_t)blockIdx.x * TPB + threadIdx.x; t < total; t += (int64_t)gridDim.x * TPB) {
        const int64_t i = t / K;
        const int k = (int)(t - i * K);
        const int64_t e = i * n_cols_logical + head_idx[k];
        const float v = Pin[i * ld + k];
        Pout[i * ld + k] = (v != 0.f && philox_uniform_at(seed, call_quads, e) < keep) ? v * scale : 0.f;
    }
}

__global__ void counter_add_kernel(int64_t* c, int64_t delta) { *c += delta; }

// ---- column sums: pass 1 = per row-chunk partials, pass 2 = partials added in chunk order ----------
__global__ __launch_bounds__(TPB) void colsum_partial_kernel(int64_t n, int F, const float* __restrict__ X,
                                                             int64_t ldx, int64_t rows_per_block,
                                                             float* __restrict__ P) {
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = min(n, r0 + rows_per_block);
    for (int col = threadIdx.x; col < F; col += TPB) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int64_t r = r0;
        for (; r + 4 <= r1; r += 4) {
            a0 += X[(r + 0) * ldx + col];
            a1 += X[(r + 1) * ldx + col];
            a2 += X[(r + 2) * ldx + col];
            a3 += X[(r + 3) * ldx + col];
        }
        for (; r < r1; ++r) a0 += X[r * ldx + col];
        P[(int64_t)blockIdx.x * F + col] = (a0 + a1) + (a2 + a3);
    }
}
// 16 columns x 16 part-groups per block: group g adds parts g, g+16, ... in order, then the 16 group
// sums are added in group order (fixed tree => deterministic)
__global__ __launch_bounds__(TPB) void colsum_final_kernel(int nparts, int F, const float* __restrict__ P,
                                                           int64_t stride, float* __restrict__ out) {
    __shared__ float s[16][17];
    const int c = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int col = blockIdx.x * 16 + c;
    float a = 0.f;
    if (col < F) {
        int p = g;                       // (four loads in flight, same order of additions)
        for (; p + 48 < nparts; p += 64) {
            const float p0 = P[(int64_t)p * stride + col], p1 = P[(int64_t)(p + 16) * stride + col];
            const float p2 = P[(int64_t)(p + 32) * stride + col], p3 = P[(int64_t)(p + 48) * stride + col];
            a = (((a + p0) + p1) + p2) + p3;
        }
        for (; p < nparts; p += 16) a += P[(int64_t)p * stride + col];
    }
    s[g][c] = a;
    __syncthreads();
    if (g == 0 && col < F) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += s[i][c];
        out[col] = t;
    }
}

__global__ __launch_bounds__(TPB) void gather_rows_kernel(int F, const float* __restrict__ X, int64_t ldx,
                                                          const int* __restrict__ idx, int64_t n_idx,
                                                          float* __restrict__ out, int64_t ldo) {
    const int64_t total = n_idx * F;
    for (int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x; e < total; e += (int64_t)gridDim.x * TPB) {
        const int64_t j = e / F;
        const int col = (int)(e - j * F);
        out[j * ldo + col] = X[(int64_t)idx[j] * ldx + col];
    }
}

// 16 bytes per lane (the halo exchange packs whole pitched rows with this: F = the row pitch)
__global__ __launch_bounds__(TPB) void gather_rows4_kernel(int F4, const float4* __restrict__ X, int64_t ldx4,
                                                           const int* __restrict__ idx, int64_t n_idx,
                                                           float4* __restrict__ out, int64_t ldo4) {
    const int64_t total = n_idx * F4;
    for (int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x; e < total; e += (int64_t)gridDim.x * TPB) {
        const int64_t j = e / F4;
        const int col = (int)(e - j * F4);
        out[j * ldo4 + col] = X[(int64_t)idx[j] * ldx4 + col];
    }
}

__global__ __launch_bounds__(TPB) void add_inplace_kernel(int64_t total4, const float4* __restrict__ X,
                                                          float4* __restrict__ Y) {
    for (int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x; e < total4; e += (int64_t)gridDim.x * TPB) {
        const float4 x = X[e];
        float4 y = Y[e];
        y.x += x.x; y.y += x.y; y.z += x.z; y.w += x.w;
        Y[e] = y;
    }
}

__global__ __launch_bounds__(TPB) void scatter_rows_kernel(int F, const float* __restrict__ S, int64_t lds,
                                                           const int* __restrict__ idx, int64_t n_idx,
                                                           float* __restrict__ out, int64_t ldo) {
    const int64_t total = n_idx * F;
    for (int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x; e < total; e += (int64_t)gridDim.x * TPB) {
        const int64_t j = e / F;
        const int col = (int)(e - j * F);
        out[(int64_t)idx[j] * ldo + col] = S[j * lds + col];
    }
}

__global__ __launch_bounds__(TPB) void pack_panels_kernel(int64_t n_rows, int64_t R, int F4, const float* __restrict__ X,
                                                          int64_t ldx, int W, int wp4, float4* __restrict__ out) {
    const int64_t total = (int64_t)W * R * wp4;
    for (int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x; e < total; e += (int64_t)gridDim.x * TPB) {
        const int j4 = (int)(e % wp4);
        const int64_t t = e / wp4;
        const int64_t i = t % R;
        const int q = (int)(t / R);
        const int c4 = q * wp4 + j4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < n_rows && c4 < F4) v = *reinterpret_cast<const float4*>(X + i * ldx + (int64_t)c4 * 4);
        out[e] = v;
    }
}

__global__ __launch_bounds__(TPB) void unpack_panels_kernel(int64_t n_rows, int64_t R, int F, int F4, const float4* __restrict__ in,
                                                            int W, int wp4, float* __restrict__ Y, int64_t ldy) {
    const int64_t total = n_rows * F4;
    for (int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x; e < total; e += (int64_t)gridDim.x * TPB) {
        const int64_t i = e / F4;
        const int c4 = (int)(e - i * F4);
        const int q = c4 / wp4, j4 = c4 - q * wp4;
        float4 v = in[((int64_t)q * R + i) * wp4 + j4];
        *reinterpret_cast<float4*>(Y + i * ldy + (int64_t)c4 * 4) = mask_pad(v, c4 * 4, F);
    }
}

int64_t colsum_parts(int64_t n) { return std::max<int64_t>(1, std::min<int64_t>(1024, cdiv(n, 64))); }

}  // namespace

int colsum_final_launch(int nparts, int F, const float* P, int64_t stride, float* out, hipStream_t st) {
    hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)cdiv(F, 16)), dim3(TPB), 0, st, nparts, F, P, stride, out);
    GEOGCN_LAUNCH_CHECK("colsum_final_kernel");
    return 0;
}

}  // namespace geogcn

using namespace geogcn;

#define CHECK_VEC(name, ld, ...)                                                                      \
    do {                                                                                              \
        const void* ptrs__[] = {__VA_ARGS__};                                                         \
        for (const void* p__ : ptrs__)                                                                \
            GEOGCN_REQUIRE(p__ && aligned16(p__), p__ ? GEOGCN_E_ALIGN : GEOGCN_E_NULL,               \
                           name ": null or misaligned pointer");                                      \
        GEOGCN_REQUIRE((ld) % 4 == 0 && (ld) >= (int64_t)((F + 3) / 4) * 4, GEOGCN_E_ALIGN,           \
                       name ": ld=%lld must be a multiple of 4 and >= roundup4(F=%d)", (long long)(ld), F); \
    } while (0)

extern "C" {

int geogcn_bias_act_f32(int64_t n, int32_t F, const float* X, int64_t ldx, const float* bias, int32_t act,
                        float* Y, int64_t ldy, void* stream) {
    GEOGCN_REQUIRE(n >= 0 && F >= 0, GEOGCN_E_SIZE, "bias_act_f32: negative size");
    if (n == 0 || F == 0) return 0;
    CHECK_VEC("bias_act_f32", ldx, X, Y);
    CHECK_VEC("bias_act_f32", ldy, X, Y);
    const int F4 = (F + 3) / 4;
    const dim3 grid(stream_grid(n * F4));
    hipStream_t st = (hipStream_t)stream;
    if (act == GEOGCN_ACT_TANH)
        hipLaunchKernelGGL((bias_act_kernel<GEOGCN_ACT_TANH>), grid, dim3(TPB), 0, st, n, F, F4, X, ldx, bias, Y, ldy);
    else if (act == GEOGCN_ACT_SIGMOID)
        hipLaunchKernelGGL((bias_act_kernel<GEOGCN_ACT_SIGMOID>), grid, dim3(TPB), 0, st, n, F, F4, X, ldx, bias, Y, ldy);
    else if (act == GEOGCN_ACT_NONE)
        hipLaunchKernelGGL((bias_act_kernel<GEOGCN_ACT_NONE>), grid, dim3(TPB), 0, st, n, F, F4, X, ldx, bias, Y, ldy);
    else if (act == GEOGCN_ACT_SELU)
        hipLaunchKernelGGL((bias_act_kernel<GEOGCN_ACT_SELU>), grid, dim3(TPB), 0, st, n, F, F4, X, ldx, bias, Y, ldy);
    else if (act == GEOGCN_ACT_RELU)
        hipLaunchKernelGGL((bias_act_kernel<GEOGCN_ACT_RELU>), grid, dim3(TPB), 0, st, n, F, F4, X, ldx, bias, Y, ldy);
    else {
        set_error("bias_act_f32: unknown act %d", act);
        return GEOGCN_E_ARG;
    }
    GEOGCN_LAUNCH_CHECK("bias_act_kernel");
    return 0;
}

int geogcn_cast_bf16_f32(int64_t n, int32_t F, const float* X, int64_t ldx, uint16_t* Y, int64_t ldy, void* stream) {
    GEOGCN_REQUIRE(n >= 0 && F >= 0, GEOGCN_E_SIZE, "cast_bf16_f32: negative size");
    if (n == 0 || F == 0) return 0;
    GEOGCN_REQUIRE(X && Y, GEOGCN_E_NULL, "cast_bf16_f32: null pointer");
    const int64_t F4 = (F + 3) / 4 * 4;
    GEOGCN_REQUIRE(ldx % 4 == 0 && ldy % 4 == 0 && ldx >= F4 && ldy >= F4 && aligned16(X) && aligned16(Y), GEOGCN_E_ALIGN,
                   "cast_bf16_f32: needs 16-byte aligned bases and ld %% 4 == 0, >= roundup4(F) (ldx=%lld ldy=%lld)",
                   (long long)ldx, (long long)ldy);
    const int Y4 = (int)(ldy / 4);
    hipLaunchKernelGGL(cast_bf16_kernel, dim3(stream_grid(n * Y4)), dim3(TPB), 0, (hipStream_t)stream, n, F, Y4, X, ldx, Y,
                       ldy);
    GEOGCN_LAUNCH_CHECK("cast_bf16_kernel");
    return 0;
}

int geogcn_highway_fwd_f32(int64_t n, int32_t F, const float* T, const float* Hc, const float* H, int64_t ld,
                           float* Hout, void* stream) {
    GEOGCN_REQUIRE(n >= 0 && F >= 0, GEOGCN_E_SIZE, "highway_fwd_f32: negative size");
    if (n == 0 || F == 0) return 0;
    CHECK_VEC("highway_fwd_f32", ld, T, Hc, H, Hout);
    // operands share one pitch and keep zero pads, so the matrices are processed as flat arrays
    const int64_t total4 = n * ld / 4;
    hipLaunchKernelGGL(highway_fwd_kernel, dim3(stream_grid(total4)), dim3(TPB), 0, (hipStream_t)stream, total4,
                       (const float4*)T, (const float4*)Hc, (const float4*)H, (float4*)Hout);
    GEOGCN_LAUNCH_CHECK("highway_fwd_kernel");
    return 0;
}

static int64_t hw_parts(int64_t n) {
    constexpr int64_t kCap = 768;      // row slices of the fused column sums (3 per CU)
    return std::max<int64_t>(1, std::min<int64_t>(kCap, cdiv(n, 64)));
}

size_t geogcn_highway_bwd_workspace_bytes(int64_t n, int32_t F) {
    if (n <= 0 || F <= 0) return 0;
    return (size_t)hw_parts(n) * 2 * (size_t)((F + 3) / 4) * 4 * sizeof(float);
}

static int highway_bwd_impl(bool s16, int64_t n, int32_t F, const float* G, const float* T, const float* Hc, const float* H,
                            int64_t ld, void* dS, int64_t ld_dS, float* dU, float* dHcarry, float* dbS, float* dbU, void* ws,
                            size_t ws_bytes, void* stream);

int geogcn_highway_bwd_f32(int64_t n, int32_t F, const float* G, const float* T, const float* Hc, const float* H,
                           int64_t ld, float* dS, int64_t ld_dS, float* dU, float* dHcarry, float* dbS, float* dbU,
                           void* ws, size_t ws_bytes, void* stream) {
    return highway_bwd_impl(false, n, F, G, T, Hc, H, ld, dS, ld_dS, dU, dHcarry, dbS, dbU, ws, ws_bytes, stream);
}

int geogcn_highway_bwd_bf16s_f32(int64_t n, int32_t F, const float* G, const float* T, const float* Hc, const float* H,
                                 int64_t ld, uint16_t* dS16, int64_t ld_dS16, float* dU, float* dHcarry, float* dbS, float* dbU,
                                 void* ws, size_t ws_bytes, void* stream) {
    GEOGCN_REQUIRE(F <= 4 * TPB, GEOGCN_E_SIZE, "highway_bwd_bf16s_f32: F=%d is wider than the fused column sums handle", F);
    return highway_bwd_impl(true, n, F, G, T, Hc, H, ld, dS16, ld_dS16, dU, dHcarry, dbS, dbU, ws, ws_bytes, stream);
}

static int highway_bwd_impl(bool s16, int64_t n, int32_t F, const float* G, const float* T, const float* Hc, const float* H,
                            int64_t ld, void* dS, int64_t ld_dS, float* dU, float* dHcarry, float* dbS, float* dbU, void* ws,
                            size_t ws_bytes, void* stream) {
    GEOGCN_REQUIRE(n >= 0 && F >= 0, GEOGCN_E_SIZE, "highway_bwd_f32: negative size");
    if (F == 0) return 0;
    if (n == 0) {
        // a rank that owns no rows: its share of the bias gradients is the zero vector (not last step's values)
        if (dbS) { const int rc = zero_fill_async(dbS, (size_t)((F + 3) / 4) * 16, (hipStream_t)stream); if (rc) return rc; }
        if (dbU) { const int rc = zero_fill_async(dbU, (size_t)((F + 3) / 4) * 16, (hipStream_t)stream); if (rc) return rc; }
        return 0;
    }
    CHECK_VEC("highway_bwd_f32", ld, G, T, Hc, H, dU);
    GEOGCN_REQUIRE(aligned16(dHcarry), GEOGCN_E_ALIGN, "highway_bwd_f32: misaligned dHcarry");          // (NULL: not stored)
    GEOGCN_REQUIRE(dS && (uintptr_t)dS % 16 == 0, GEOGCN_E_NULL, "highway_bwd_f32: dS is null or not 16-byte aligned");
    GEOGCN_REQUIRE(ld_dS % (s16 ? 8 : 4) == 0 && ld_dS >= ld, GEOGCN_E_ALIGN,
                   "highway_bwd_f32: ld_dS=%lld must be a multiple of %d, >= ld", (long long)ld_dS, s16 ? 8 : 4);
    GEOGCN_REQUIRE((dbS == nullptr) == (dbU == nullptr), GEOGCN_E_ARG, "highway_bwd_f32: pass both bias gradients or neither");
    hipStream_t st = (hipStream_t)stream;
    const int ld4 = (int)(ld / 4);
    if (dbS && ld4 <= TPB && ld == (int64_t)((F + 3) / 4) * 4) {
        const int64_t parts = hw_parts(n);
        const int64_t rpb = cdiv(n, parts);
        const int nparts = (int)cdiv(n, rpb);
        GEOGCN_REQUIRE(ws && aligned16(ws) && ws_bytes >= (size_t)nparts * 2 * ld * sizeof(float), GEOGCN_E_ARG,
                       "highway_bwd_f32: workspace too small");
        if (s16)
            hipLaunchKernelGGL(highway_bwd_colsum_kernel<true>, dim3((unsigned)nparts), dim3(TPB), 0, st, n, ld4, (const float4*)G,
                               (const float4*)T, (const float4*)Hc, (const float4*)H, dS, (int)(ld_dS / 4), (float4*)dU,
                               (float4*)dHcarry, rpb, (float4*)ws);
        else
            hipLaunchKernelGGL(highway_bwd_colsum_kernel<false>, dim3((unsigned)nparts), dim3(TPB), 0, st, n, ld4, (const float4*)G,
                               (const float4*)T, (const float4*)Hc, (const float4*)H, dS, (int)(ld_dS / 4), (float4*)dU,
                               (float4*)dHcarry, rpb, (float4*)ws);
        GEOGCN_LAUNCH_CHECK("highway_bwd_colsum_kernel");
        hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)cdiv(F, 16)), dim3(TPB), 0, st, nparts, F, (const float*)ws,
                           (int64_t)2 * ld, dbS);
        hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)cdiv(F, 16)), dim3(TPB), 0, st, nparts, F,
                           (const float*)ws + ld, (int64_t)2 * ld, dbU);
        GEOGCN_LAUNCH_CHECK("colsum_final_kernel");
        return 0;
    }
    GEOGCN_REQUIRE(!(s16 && dbS), GEOGCN_E_ARG, "highway_bwd_bf16s_f32: bias gradients need ld == roundup4(F) <= %d", 4 * TPB);
    if (s16)
        hipLaunchKernelGGL(highway_bwd_kernel<true>, dim3(stream_grid(n * ld / 4)), dim3(TPB), 0, st, n, ld4, (const float4*)G,
                           (const float4*)T, (const float4*)Hc, (const float4*)H, dS, (int)(ld_dS / 4), (float4*)dU, (float4*)dHcarry);
    else
        hipLaunchKernelGGL(highway_bwd_kernel<false>, dim3(stream_grid(n * ld / 4)), dim3(TPB), 0, st, n, ld4, (const float4*)G,
                           (const float4*)T, (const float4*)Hc, (const float4*)H, dS, (int)(ld_dS / 4), (float4*)dU, (float4*)dHcarry);
    GEOGCN_LAUNCH_CHECK("highway_bwd_kernel");
    if (dbS) {          // very wide layers: separate deterministic column sums
        int rc = geogcn_colsum_f32(n, F, (const float*)dS, ld_dS, dbS, ws, ws_bytes, stream);
        if (rc) return rc;
        return geogcn_colsum_f32(n, F, dU, ld, dbU, ws, ws_bytes, stream);
    }
    return 0;
}

int geogcn_act_bwd_f32(int64_t n, int32_t F, const float* G, const float* Y, int64_t ld, int32_t act,
                       const uint8_t* keep_mask, float scale, float* dS, int64_t ld_dS, void* stream) {
    GEOGCN_REQUIRE(n >= 0 && F >= 0, GEOGCN_E_SIZE, "act_bwd_f32: negative size");
    if (n == 0 || F == 0) return 0;
    CHECK_VEC("act_bwd_f32", ld, G, Y, dS);
    GEOGCN_REQUIRE(ld_dS % 4 == 0 && ld_dS >= (int64_t)((F + 3) / 4) * 4, GEOGCN_E_ALIGN, "act_bwd_f32: bad ld_dS=%lld",
                   (long long)ld_dS);
    const int F4 = (F + 3) / 4;
    const dim3 grid(stream_grid(n * F4));
    hipStream_t st = (hipStream_t)stream;
    if (act == GEOGCN_ACT_TANH)
        hipLaunchKernelGGL((act_bwd_kernel<GEOGCN_ACT_TANH>), grid, dim3(TPB), 0, st, n, F, F4, G, Y, ld, keep_mask, scale, dS, ld_dS);
    else if (act == GEOGCN_ACT_SIGMOID)
        hipLaunchKernelGGL((act_bwd_kernel<GEOGCN_ACT_SIGMOID>), grid, dim3(TPB), 0, st, n, F, F4, G, Y, ld, keep_mask, scale, dS, ld_dS);
    else if (act == GEOGCN_ACT_NONE)
        hipLaunchKernelGGL((act_bwd_kernel<GEOGCN_ACT_NONE>), grid, dim3(TPB), 0, st, n, F, F4, G, Y, ld, keep_mask, scale, dS, ld_dS);
    else if (act == GEOGCN_ACT_SELU)
        hipLaunchKernelGGL((act_bwd_kernel<GEOGCN_ACT_SELU>), grid, dim3(TPB), 0, st, n, F, F4, G, Y, ld, keep_mask, scale, dS, ld_dS);
    else if (act == GEOGCN_ACT_RELU)
        hipLaunchKernelGGL((act_bwd_kernel<GEOGCN_ACT_RELU>), grid, dim3(TPB), 0, st, n, F, F4, G, Y, ld, keep_mask, scale, dS, ld_dS);
    else {
        set_error("act_bwd_f32: unknown act %d", act);
        return GEOGCN_E_ARG;
    }
    GEOGCN_LAUNCH_CHECK("act_bwd_kernel");
    return 0;
}

size_t geogcn_colsum_workspace_bytes(int64_t n, int32_t F) {
    if (n <= 0 || F <= 0) return 0;
    return (size_t)colsum_parts(n) * (size_t)F * sizeof(float);
}

int geogcn_colsum_f32(int64_t n, int32_t F, const float* X, int64_t ldx, float* out, void* ws, size_t ws_bytes,
                      void* stream) {
    GEOGCN_REQUIRE(n >= 0 && F >= 0, GEOGCN_E_SIZE, "colsum_f32: negative size");
    if (F == 0) return 0;
    GEOGCN_REQUIRE(out, GEOGCN_E_NULL, "colsum_f32: null out");
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) {
        { const int zrc = zero_fill_async(out, (size_t)F * sizeof(float), st); if (zrc) return zrc; }
        return 0;
    }
    GEOGCN_REQUIRE(X && ldx >= F, GEOGCN_E_NULL, "colsum_f32: null X or ldx < F");
    const int64_t parts = colsum_parts(n);
    GEOGCN_REQUIRE(ws && ws_bytes >= (size_t)parts * F * sizeof(float), GEOGCN_E_ARG, "colsum_f32: workspace too small");
    const int64_t rpb = cdiv(n, parts);
    const int nparts = (int)cdiv(n, rpb);
    hipLaunchKernelGGL(colsum_partial_kernel, dim3((unsigned)nparts), dim3(TPB), 0, st, n, F, X, ldx, rpb, (float*)ws);
    GEOGCN_LAUNCH_CHECK("colsum_partial_kernel");
    hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)cdiv(F, 16)), dim3(TPB), 0, st, nparts, F,
                       (const float*)ws, (int64_t)F, out);
    GEOGCN_LAUNCH_CHECK("colsum_final_kernel");
    return 0;
}

int geogcn_dropout_mask_philox(int64_t n, int32_t F, float p_drop, uint64_t seed, uint64_t offset,
                               uint8_t* keep_mask, void* stream) {
    GEOGCN_REQUIRE(n >= 0 && F >= 0, GEOGCN_E_SIZE, "dropout_mask_philox: negative size");
    GEOGCN_REQUIRE(p_drop >= 0.f && p_drop < 1.f, GEOGCN_E_ARG, "dropout_mask_philox: p=%f outside [0,1)", p_drop);
    if (n == 0 || F == 0) return 0;
    GEOGCN_REQUIRE(keep_mask, GEOGCN_E_NULL, "dropout_mask_philox: null mask");
    const int64_t total = n * F;
    hipLaunchKernelGGL(dropout_mask_kernel, dim3(stream_grid((total + 3) / 4)), dim3(TPB), 0, (hipStream_t)stream,
                       total, 1.0f - p_drop, seed, offset, (const int64_t*)nullptr, (int64_t)0, (int64_t)0, keep_mask);
    GEOGCN_LAUNCH_CHECK("dropout_mask_kernel");
    return 0;
}

int geogcn_dropout_mask_philox_ctr(int64_t n, int32_t F, float p_drop, uint64_t seed, const int64_t* calls_dev,
                                   int64_t per_call_elems, int64_t base_elems, uint8_t* keep_mask, void* stream) {
    GEOGCN_REQUIRE(n >= 0 && F >= 0 && per_call_elems >= 0 && base_elems >= 0, GEOGCN_E_SIZE,
                   "dropout_mask_philox_ctr: negative size");
    GEOGCN_REQUIRE(p_drop >= 0.f && p_drop < 1.f, GEOGCN_E_ARG, "dropout_mask_philox_ctr: p=%f outside [0,1)", p_drop);
    if (n == 0 || F == 0) return 0;
    GEOGCN_REQUIRE(keep_mask && calls_dev, GEOGCN_E_NULL, "dropout_mask_philox_ctr: null pointer");
    const int64_t total = n * F;
    hipLaunchKernelGGL(dropout_mask_kernel, dim3(stream_grid((total + 3) / 4)), dim3(TPB), 0, (hipStream_t)stream,
                       total, 1.0f - p_drop, seed, (uint64_t)0, calls_dev, per_call_elems, base_elems, keep_mask);
    GEOGCN_LAUNCH_CHECK("dropout_mask_kernel");
    return 0;
}

int geogcn_counter_add_i64(int64_t* counter_dev, int64_t delta, void* stream) {
    GEOGCN_REQUIRE(counter_dev, GEOGCN_E_NULL, "counter_add_i64: null pointer");
    hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, counter_dev, delta);
    GEOGCN_LAUNCH_CHECK("counter_add_kernel");
    return 0;
}

int geogcn_dropout_apply_f32(int64_t n, int32_t F, const float* X, int64_t ld, const uint8_t* keep_mask,
                             float p_drop, float* Y, void* stream) {
    GEOGCN_REQUIRE(n >= 0 && F >= 0, GEOGCN_E_SIZE, "dropout_apply_f32: negative size");
    GEOGCN_REQUIRE(p_drop >= 0.f && p_drop < 1.f, GEOGCN_E_ARG, "dropout_apply_f32: p=%f outside [0,1)", p_drop);
    if (n == 0 || F == 0) return 0;
    GEOGCN_REQUIRE(keep_mask, GEOGCN_E_NULL, "dropout_apply_f32: null mask");
    CHECK_VEC("dropout_apply_f32", ld, X, Y);
    const int F4 = (F + 3) / 4;
    hipLaunchKernelGGL(dropout_apply_kernel, dim3(stream_grid(n * F4)), dim3(TPB), 0, (hipStream_t)stream, n, F, F4, X,
                       ld, keep_mask, 1.0f / (1.0f - p_drop), Y);
    GEOGCN_LAUNCH_CHECK("dropout_apply_kernel");
    return 0;
}

int geogcn_scatter_rows_f32(int32_t F, const float* src, int64_t lds, const int32_t* idx, int64_t n_idx, float* out,
                            int64_t ldo, void* stream) {
    GEOGCN_REQUIRE(F >= 0 && n_idx >= 0, GEOGCN_E_SIZE, "scatter_rows_f32: negative size");
    if (F == 0 || n_idx == 0) return 0;
    GEOGCN_REQUIRE(src && idx && out, GEOGCN_E_NULL, "scatter_rows_f32: null pointer");
    GEOGCN_REQUIRE(lds >= F && ldo >= F, GEOGCN_E_SIZE, "scatter_rows_f32: ld < F");
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(stream_grid(n_idx * F)), dim3(TPB), 0, (hipStream_t)stream, F, src, lds,
                       idx, n_idx, out, ldo);
    GEOGCN_LAUNCH_CHECK("scatter_rows_kernel");
    return 0;
}

int geogcn_pack_panels_f32(int64_t n_rows, int64_t R, int32_t F, const float* X, int64_t ldx, int32_t W, int32_t wp,
                           float* out, void* stream) {
    GEOGCN_REQUIRE(n_rows >= 0 && R >= n_rows && F > 0 && W > 0 && wp > 0, GEOGCN_E_SIZE, "pack_panels_f32: bad sizes");
    GEOGCN_REQUIRE(wp % 4 == 0 && (int64_t)W * wp >= F && ldx % 4 == 0 && ldx >= (int64_t)((F + 3) / 4) * 4, GEOGCN_E_ALIGN,
                   "pack_panels_f32: need wp %% 4 == 0, W*wp >= F, ldx %% 4 == 0");
    if (R == 0) return 0;
    // (a rank that owns no rows still zero-fills its R padded panel rows; its X is an empty matrix whose pointer may be null)
    GEOGCN_REQUIRE((X || n_rows == 0) && out && aligned16(X) && aligned16(out), GEOGCN_E_NULL, "pack_panels_f32: null/misaligned pointer");
    const int64_t total = (int64_t)W * R * (wp / 4);
    hipLaunchKernelGGL(pack_panels_kernel, dim3(stream_grid(total)), dim3(TPB), 0, (hipStream_t)stream, n_rows, R, (F + 3) / 4,
                       X, ldx, W, wp / 4, (float4*)out);
    GEOGCN_LAUNCH_CHECK("pack_panels_kernel");
    return 0;
}

int geogcn_unpack_panels_f32(int64_t n_rows, int64_t R, int32_t F, const float* in, int32_t W, int32_t wp, float* Y,
                             int64_t ldy, void* stream) {
    GEOGCN_REQUIRE(n_rows >= 0 && R >= n_rows && F > 0 && W > 0 && wp > 0, GEOGCN_E_SIZE, "unpack_panels_f32: bad sizes");
    GEOGCN_REQUIRE(wp % 4 == 0 && (int64_t)W * wp >= F && ldy % 4 == 0 && ldy >= (int64_t)((F + 3) / 4) * 4, GEOGCN_E_ALIGN,
                   "unpack_panels_f32: need wp %% 4 == 0, W*wp >= F, ldy %% 4 == 0");
    if (n_rows == 0) return 0;
    GEOGCN_REQUIRE(in && Y && aligned16(in) && aligned16(Y), GEOGCN_E_NULL, "unpack_panels_f32: null/misaligned pointer");
    const int F4 = (F + 3) / 4;
    hipLaunchKernelGGL(unpack_panels_kernel, dim3(stream_grid(n_rows * F4)), dim3(TPB), 0, (hipStream_t)stream, n_rows, R, F, F4,
                       (const float4*)in, W, wp / 4, Y, ldy);
    GEOGCN_LAUNCH_CHECK("unpack_panels_kernel");
    return 0;
}

int geogcn_gate_carry_f32(int64_t n, int32_t F, const float* G, int64_t ldg, const float* T, int64_t ldt, float* out,
                          int64_t ldo, void* stream) {
    GEOGCN_REQUIRE(n >= 0 && F >= 0, GEOGCN_E_SIZE, "gate_carry_f32: negative size");
    if (n == 0 || F == 0) return 0;
    GEOGCN_REQUIRE(G && T && out, GEOGCN_E_NULL, "gate_carry_f32: null pointer");
    const int64_t f4 = ((int64_t)F + 3) / 4 * 4;
    GEOGCN_REQUIRE(aligned16(G) && aligned16(T) && aligned16(out) && ldg % 4 == 0 && ldt % 4 == 0 && ldo % 4 == 0 && ldg >= f4 &&
                       ldt >= f4 && ldo >= f4,
                   GEOGCN_E_ALIGN, "gate_carry_f32: needs 16-byte aligned operands and pitches that are multiples of 4, >= roundup4(F)");
    hipLaunchKernelGGL(gate_carry_kernel, dim3(stream_grid(n * (f4 / 4))), dim3(TPB), 0, (hipStream_t)stream, n, (int)(f4 / 4), G, ldg,
                       T, ldt, out, ldo);
    GEOGCN_LAUNCH_CHECK("gate_carry_kernel");
    return 0;
}

int geogcn_add_inplace_f32(int64_t n_floats, const float* X, float* Y, void* stream) {
    GEOGCN_REQUIRE(n_floats >= 0, GEOGCN_E_SIZE, "add_inplace_f32: negative size");
    if (n_floats == 0) return 0;
    GEOGCN_REQUIRE(X && Y, GEOGCN_E_NULL, "add_inplace_f32: null pointer");
    GEOGCN_REQUIRE(n_floats % 4 == 0 && aligned16(X) && aligned16(Y), GEOGCN_E_ALIGN,
                   "add_inplace_f32: needs 16-byte aligned operands and n %% 4 == 0");
    hipLaunchKernelGGL(add_inplace_kernel, dim3(stream_grid(n_floats / 4)), dim3(TPB), 0, (hipStream_t)stream,
                       n_floats / 4, (const float4*)X, (float4*)Y);
    GEOGCN_LAUNCH_CHECK("add_inplace_kernel");
    return 0;
}

int geogcn_gather_rows_f32(int32_t F, const float* X, int64_t ldx, const int32_t* idx, int64_t n_idx, float* out,
                           int64_t ldo, void* stream) {
    GEOGCN_REQUIRE(F >= 0 && n_idx >= 0, GEOGCN_E_SIZE, "gather_rows_f32: negative size");
    if (F == 0 || n_idx == 0) return 0;
    GEOGCN_REQUIRE(X && idx && out, GEOGCN_E_NULL, "gather_rows_f32: null pointer");
    GEOGCN_REQUIRE(ldx >= F && ldo >= F, GEOGCN_E_SIZE, "gather_rows_f32: ld < F");
    if (F % 4 == 0 && ldx % 4 == 0 && ldo % 4 == 0 && ((uintptr_t)X | (uintptr_t)out) % 16 == 0) {
        hipLaunchKernelGGL(gather_rows4_kernel, dim3(stream_grid(n_idx * (F / 4))), dim3(TPB), 0, (hipStream_t)stream, F / 4,
                           (const float4*)X, ldx / 4, idx, n_idx, (float4*)out, ldo / 4);
        GEOGCN_LAUNCH_CHECK("gather_rows4_kernel");
        return 0;
    }
    hipLaunchKernelGGL(gather_rows_kernel, dim3(stream_grid(n_idx * F)), dim3(TPB), 0, (hipStream_t)stream, F, X, ldx,
                       idx, n_idx, out, ldo);
    GEOGCN_LAUNCH_CHECK("gather_rows_kernel");
    return 0;
}

int geogcn_act_bwd_colsum_f32(int64_t n, int32_t F, const float* G, const float* Y, int64_t ld, int32_t act,
                              const uint8_t* keep_mask, float scale, float* dS, int64_t ld_dS, float* db, void* ws,
                              size_t ws_bytes, void* stream) {
    GEOGCN_REQUIRE(n >= 0 && F >= 0, GEOGCN_E_SIZE, "act_bwd_colsum_f32: negative size");
    GEOGCN_REQUIRE(db, GEOGCN_E_NULL, "act_bwd_colsum_f32: null db");
    if (F == 0) return 0;
    if (n == 0) return zero_fill_async(db, (size_t)((F + 3) / 4) * 16, (hipStream_t)stream);      // empty row share: db = 0
    const int64_t F4 = (F + 3) / 4;
    const bool fused = (act == GEOGCN_ACT_TANH || act == GEOGCN_ACT_SIGMOID || act == GEOGCN_ACT_NONE) && ld == F4 * 4 &&
                       F4 <= TPB;
    if (!fused) {          // odd pitch / other activations: two passes, same results
        int rc = geogcn_act_bwd_f32(n, F, G, Y, ld, act, keep_mask, scale, dS, ld_dS, stream);
        if (rc) return rc;
        return geogcn_colsum_f32(n, F, dS, ld_dS, db, ws, ws_bytes, stream);
    }
    CHECK_VEC("act_bwd_colsum_f32", ld, G, Y, dS);
    GEOGCN_REQUIRE(ld_dS % 4 == 0 && ld_dS >= ld, GEOGCN_E_ALIGN, "act_bwd_colsum_f32: bad ld_dS=%lld", (long long)ld_dS);
    const int64_t parts = hw_parts(n);
    const int64_t rpb = cdiv(n, parts);
    const int nparts = (int)cdiv(n, rpb);
    GEOGCN_REQUIRE(ws && aligned16(ws) && ws_bytes >= (size_t)nparts * ld * sizeof(float), GEOGCN_E_ARG,
                   "act_bwd_colsum_f32: workspace too small (see geogcn_highway_bwd_workspace_bytes)");
    hipStream_t st = (hipStream_t)stream;
    const int ld4 = (int)(ld / 4);
#define GEOGCN_AB(ACT)                                                                                               \
    hipLaunchKernelGGL((act_bwd_colsum_kernel<ACT>), dim3((unsigned)nparts), dim3(TPB), 0, st, n, F, ld4, (const float4*)G, \
                       (const float4*)Y, keep_mask, scale, (float4*)dS, (int)(ld_dS / 4), rpb, (float4*)ws)
    if (act == GEOGCN_ACT_TANH) GEOGCN_AB(GEOGCN_ACT_TANH);
    else if (act == GEOGCN_ACT_SIGMOID) GEOGCN_AB(GEOGCN_ACT_SIGMOID);
    else GEOGCN_AB(GEOGCN_ACT_NONE);
#undef GEOGCN_AB
    GEOGCN_LAUNCH_CHECK("act_bwd_colsum_kernel");
    hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)cdiv(F, 16)), dim3(TPB), 0, st, nparts, F, (const float*)ws, ld, db);
    GEOGCN_LAUNCH_CHECK("colsum_final_kernel");
    return 0;
}

int geogcn_colsum_rowblocks_f32(int64_t n, int32_t F, const float* X, int64_t ldx, float* out, void* ws, size_t ws_bytes, void* stream) {
    GEOGCN_REQUIRE(n >= 0 && F >= 0, GEOGCN_E_SIZE, "colsum_rowblocks_f32: negative size");
    GEOGCN_REQUIRE(out, GEOGCN_E_NULL, "colsum_rowblocks_f32: null out");
    if (F == 0) return 0;
    if (n == 0) return zero_fill_async(out, (size_t)((F + 3) / 4) * 16, (hipStream_t)stream);
    const int64_t F4 = (F + 3) / 4;
    GEOGCN_REQUIRE(X && aligned16(X) && ldx % 4 == 0 && ldx >= F4 * 4, GEOGCN_E_ALIGN,
                   "colsum_rowblocks_f32: X needs a 16-byte aligned base and a pitch %% 4 == 0, >= roundup4(F)");
    if (F4 > TPB) return geogcn_colsum_f32(n, F, X, ldx, out, ws, ws_bytes, stream);       // (wider than the fused kernels handle)
    const int64_t parts = hw_parts(n);
    const int64_t rpb = cdiv(n, parts);
    const int nparts = (int)cdiv(n, rpb);
    GEOGCN_REQUIRE(ws && aligned16(ws) && ws_bytes >= (size_t)nparts * F4 * 4 * sizeof(float), GEOGCN_E_ARG,
                   "colsum_rowblocks_f32: workspace too small (see geogcn_highway_bwd_workspace_bytes)");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(colsum_rowblocks_kernel, dim3((unsigned)nparts), dim3(TPB), 0, st, n, F, (int)F4, (const float4*)X, (int)(ldx / 4), rpb,
                       (float4*)ws);
    GEOGCN_LAUNCH_CHECK("colsum_rowblocks_kernel");
    hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)cdiv(F, 16)), dim3(TPB), 0, st, nparts, F, (const float*)ws, F4 * 4, out);
    GEOGCN_LAUNCH_CHECK("colsum_final_kernel");
    return 0;
}

int geogcn_dropout_csr_f32(int64_t n_rows, const int32_t* rowptr, const int32_t* colidx, const float* val_in, float* val_out,
                           int64_t n_rows_logical, int64_t n_cols_logical, int32_t transposed, float p_drop, uint64_t seed,
                           uint64_t call, void* stream) {
    GEOGCN_REQUIRE(n_rows >= 0 && n_rows_logical >= 0 && n_cols_logical >= 0, GEOGCN_E_SIZE, "dropout_csr_f32: negative size");
    GEOGCN_REQUIRE(p_drop >= 0.f && p_drop < 1.f, GEOGCN_E_ARG, "dropout_csr_f32: p=%f outside [0,1)", p_drop);
    if (n_rows == 0) return 0;
    GEOGCN_REQUIRE(rowptr && colidx && val_in && val_out, GEOGCN_E_NULL, "dropout_csr_f32: null pointer");
    const uint64_t quads = (uint64_t)((n_rows_logical * n_cols_logical + 3) / 4);
    hipLaunchKernelGGL(dropout_csr_kernel, dim3((unsigned)cdiv(n_rows, TPB / kWave)), dim3(TPB), 0, (hipStream_t)stream, n_rows,
                       rowptr, colidx, val_in, val_out, n_cols_logical, transposed, 1.0f - p_drop, 1.0f / (1.0f - p_drop), seed,
                       call * quads);
    GEOGCN_LAUNCH_CHECK("dropout_csr_kernel");
    return 0;
}

int geogcn_dropout_panel_f32(int64_t n, int32_t K, const float* panel_in, int64_t ld, const int32_t* head_idx,
                             int64_t n_cols_logical, float p_drop, uint64_t seed, uint64_t call, float* panel_out,
                             void* stream) {
    GEOGCN_REQUIRE(n >= 0 && K >= 0 && n_cols_logical >= 0, GEOGCN_E_SIZE, "dropout_panel_f32: negative size");
    GEOGCN_REQUIRE(p_drop >= 0.f && p_drop < 1.f, GEOGCN_E_ARG, "dropout_panel_f32: p=%f outside [0,1)", p_drop);
    if (n == 0 || K == 0) return 0;
    GEOGCN_REQUIRE(panel_in && panel_out && head_idx && ld >= K, GEOGCN_E_NULL, "dropout_panel_f32: null pointer / ld < K");
    const uint64_t quads = (uint64_t)((n * n_cols_logical + 3) / 4);
    hipLaunchKernelGGL(dropout_panel_kernel, dim3(stream_grid(n * K)), dim3(TPB), 0, (hipStream_t)stream, n, K, panel_in, ld,
                       head_idx, n_cols_logical, 1.0f - p_drop, 1.0f / (1.0f - p_drop), seed, call * quads, panel_out);
    GEOGCN_LAUNCH_CHECK("dropout_panel_kernel");
    return 0;
}

}  // extern "C"
