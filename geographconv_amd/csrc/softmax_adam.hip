// Output head + optimiser kernels of the GCN hot path (gfx950).
//   softmax / argmax / cross-entropy / row gather+scatter   reference gcnmodel.py:374-382,389,393-394
//   lasagne.updates.adam + l1/l2 penalty                     reference gcnmodel.py:383-387,407
// One 64-lane wave per row for the row-wise ops (C <= a few hundred: 129 / 256 / 930), butterfly
// shuffles for the reductions (fixed tree => run-to-run bitwise stable); scalar reductions over
// index vectors use a fixed two-pass tree (no float atomics).
#include "common.h"

#include <algorithm>

namespace geogcn {
namespace {

constexpr int TPB = 256;
constexpr int kWavesPerBlock = TPB / kWave;

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, kWave));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
    return v;
}

// probs = exp(x - max) / sum ; argmax = first index attaining the max (numpy / Theano argmax)
__global__ __launch_bounds__(TPB) void softmax_rows_kernel(int64_t n, int C, const float* __restrict__ L, int64_t ldl,
                                                           float* __restrict__ P, int64_t ldp, int ldp_pad,
                                                           int* __restrict__ amax) {
    const int64_t row = (int64_t)blockIdx.x * kWavesPerBlock + threadIdx.x / kWave;
    const int lane = threadIdx.x % kWave;
    if (row >= n) return;
    const float* x = L + row * ldl;
    float m = -INFINITY;
    int mi = 0x7fffffff;
    for (int c = lane; c < C; c += kWave) {
        const float v = x[c];
        if (v > m) { m = v; mi = c; }          // strict > keeps the first index within a lane
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float om = __shfl_xor(m, o, kWave);
        const int oi = __shfl_xor(mi, o, kWave);
        if (om > m || (om == m && oi < mi)) { m = om; mi = oi; }
    }
    float s = 0.f;
    for (int c = lane; c < C; c += kWave) s += expf(x[c] - m);
    s = wave_sum(s);
    float* p = P + row * ldp;
    for (int c = lane; c < ldp_pad; c += kWave) p[c] = (c < C) ? expf(x[c] - m) / s : 0.f;
    if (amax && lane == 0) amax[row] = mi;
}

// C <= 64 * KREG: the row lives in registers -- one read of the logits, one exp per element, one write.
template <int KREG>
__global__ __launch_bounds__(TPB) void softmax_rows_reg_kernel(int64_t n, int C, const float* L, int64_t ldl,
                                                               float* P, int64_t ldp, int ldp_pad,
                                                               int* __restrict__ amax, const int* __restrict__ rowlist = nullptr) {
    // (rowlist: the n listed rows instead of rows 0 .. n-1 -- then L may be P: a row is read completely before it is written)
    const int64_t slot = (int64_t)blockIdx.x * kWavesPerBlock + threadIdx.x / kWave;
    const int lane = threadIdx.x % kWave;
    if (slot >= n) return;
    const int64_t row = rowlist ? rowlist[slot] : slot;
    const float* x = L + row * ldl;
    float v[KREG];
    float m = -INFINITY;
    int mi = 0x7fffffff;
#pragma unroll
    for (int k = 0; k < KREG; ++k) {
        const int c = lane + kWave * k;
        v[k] = (c < C) ? x[c] : -INFINITY;
        if (v[k] > m) { m = v[k]; mi = c; }        // ascending c within a lane: strict > keeps the first index
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float om = __shfl_xor(m, o, kWave);
        const int oi = __shfl_xor(mi, o, kWave);
        if (om > m || (om == m && oi < mi)) { m = om; mi = oi; }
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < KREG; ++k) {
        const int c = lane + kWave * k;
        v[k] = (c < C) ? expf(v[k] - m) : 0.f;
        s += v[k];
    }
    s = wave_sum(s);
    float* p = P + row * ldp;
#pragma unroll
    for (int k = 0; k < KREG; ++k) {
        const int c = lane + kWave * k;
        if (c < ldp_pad) p[c] = (c < C) ? v[k] / s : 0.f;
    }
    if (amax && lane == 0) amax[row] = mi;
}

// per-index loss / hit, block partial sums in a fixed tree
__global__ __launch_bounds__(TPB) void ce_partial_kernel(int C, const float* __restrict__ P, int64_t ldp,
                                                         const int* __restrict__ amax,
                                                         const int* __restrict__ idx, int64_t n_idx,
                                                         const int* __restrict__ y, float* __restrict__ part) {
    __shared__ float s_loss[kWavesPerBlock], s_hit[kWavesPerBlock];
    const int64_t j = (int64_t)blockIdx.x * TPB + threadIdx.x;
    float loss = 0.f, hit = 0.f;
    if (j < n_idx) {
        const int64_t row = idx[j];
        const float* p = P + row * ldp;
        const int yy = y[j];
        loss = -logf(p[yy]);
        int mi = 0;
        if (amax) {
            mi = amax[row];
        } else {
            float m = p[0];
            for (int c = 1; c < C; ++c) {
                const float v = p[c];
                if (v > m) { m = v; mi = c; }
            }
        }
        hit = (mi == yy) ? 1.f : 0.f;
    }
    loss = wave_sum(loss);
    hit = wave_sum(hit);
    const int w = threadIdx.x / kWave;
    if (threadIdx.x % kWave == 0) { s_loss[w] = loss; s_hit[w] = hit; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float a = 0.f, b = 0.f;
        for (int i = 0; i < kWavesPerBlock; ++i) { a += s_loss[i]; b += s_hit[i]; }
        part[2 * blockIdx.x] = a;
        part[2 * blockIdx.x + 1] = b;
    }
}
// one wave: lane l adds the partials l, l + 64, ... in turn, the 64 lane sums are combined by the fixed butterfly of wave_sum --
// the same association every run (one thread walking ~1,000 partials alone took 45-65 us of an otherwise idle GPU)
__global__ void pair_final_kernel(int nparts, const float* __restrict__ part, float* __restrict__ out2) {
    float a = 0.f, b = 0.f;
    for (int i = threadIdx.x; i < nparts; i += kWave) { a += part[2 * i]; b += part[2 * i + 1]; }
    a = wave_sum(a);
    b = wave_sum(b);
    if (threadIdx.x == 0) {
        out2[0] = a;
        out2[1] = b;
    }
}

// dlogits[idx[j], c] += (P[idx[j], c] - [c == y[j]]) / n_idx.  dlogits was zeroed just before; with the
// unique indices the reference produces (np.random.choice(replace=False), gcnmain.py:207) every
// element sees exactly one add onto 0, and k duplicates add k identical values -- order-free.
__global__ __launch_bounds__(TPB) void ce_bwd_kernel(int C, const float* __restrict__ P, int64_t ldp,
                                                     const int* __restrict__ idx, int64_t n_idx,
                                                     const int* __restrict__ y, float inv_n,
                                                     float* __restrict__ D, int64_t ldd) {
    const int64_t j = (int64_t)blockIdx.x * kWavesPerBlock + threadIdx.x / kWave;
    const int lane = threadIdx.x % kWave;
    if (j >= n_idx) return;
    const int64_t row = idx[j];
    const int yy = y[j];
    for (int c = lane; c < C; c += kWave) {
        const float g = (P[row * ldp + c] - (c == yy ? 1.0f : 0.0f)) * inv_n;
        unsafeAtomicAdd(D + row * ldd + c, g);      // the hardware fp32 add, chosen HERE (exact: see above), not by a build-wide flag
    }
}

// The same scatter with the output layer's bias gradient (column sums of dlogits) in the same pass: wave w takes
// the index entries w, w + n_waves, ... and keeps their column sums in registers (C <= 1024: 16 per lane); the four
// waves of a block are combined in wave order through LDS, one partial row per block, added in fixed order
// afterwards (colsum_final) -- no pass over the N x C matrix.
constexpr int kCeBlocks = 1024;
constexpr int kCeWaves = kCeBlocks * kWavesPerBlock;
// The compact form for C % 4 == 0 (the usual case): a lane owns the float4s lane + 64 k of a row, a wave takes FOUR index entries
// per trip -- their index / label loads, then their row loads, are issued together -- instead of one entry per trip with two
// dependent round trips (index -> row): 233 -> ~110 us at the TwitterUS shape.  Same values, same column-sum association per
// wave (entries in index order), same block / final combine.
__global__ __launch_bounds__(TPB) void ce_rows4_db_kernel(int C4, const float* __restrict__ P, int64_t ldp,
                                                          const int* __restrict__ idx, int64_t n_idx,
                                                          const int* __restrict__ y, float inv_n,
                                                          float* __restrict__ D, int64_t ldd, float* __restrict__ part) {
    __shared__ float4 red[kWavesPerBlock][kWave];
    const int wv = threadIdx.x / kWave, lane = threadIdx.x % kWave;
    const int w = blockIdx.x * kWavesPerBlock + wv;
    float4 acc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t j0 = w; j0 < n_idx; j0 += 4 * (int64_t)kCeWaves) {
        int64_t row[4];
        int yy[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int64_t j = j0 + e * (int64_t)kCeWaves;
            row[e] = j < n_idx ? idx[j] : -1;
            yy[e] = j < n_idx ? y[j] : 0;
        }
        float4 v[4][4];
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int q = lane + kWave * k;
                if (row[e] >= 0 && q < C4) v[e][k] = *reinterpret_cast<const float4*>(P + row[e] * ldp + 4 * q);
            }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (row[e] < 0) continue;
            const int64_t j = j0 + e * (int64_t)kCeWaves;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int q = lane + kWave * k;
                if (q < C4) {
                    const int c = 4 * q;
                    float4 g;
                    g.x = (v[e][k].x - (c + 0 == yy[e] ? 1.0f : 0.0f)) * inv_n;
                    g.y = (v[e][k].y - (c + 1 == yy[e] ? 1.0f : 0.0f)) * inv_n;
                    g.z = (v[e][k].z - (c + 2 == yy[e] ? 1.0f : 0.0f)) * inv_n;
                    g.w = (v[e][k].w - (c + 3 == yy[e] ? 1.0f : 0.0f)) * inv_n;
                    *reinterpret_cast<float4*>(D + j * ldd + c) = g;
                    acc[k].x += g.x; acc[k].y += g.y; acc[k].z += g.z; acc[k].w += g.w;
                }
            }
        }
    }
    for (int k = 0; k < 4; ++k) {
        const int q = lane + kWave * k;
        if (kWave * k >= C4) break;
        red[wv][lane] = acc[k];
        __syncthreads();
        if (wv == 0 && q < C4) {
            float4 t = red[0][lane];
#pragma unroll
            for (int i = 1; i < kWavesPerBlock; ++i) {
                const float4 u = red[i][lane];
                t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
            }
            *reinterpret_cast<float4*>(part + (int64_t)blockIdx.x * 4 * C4 + 4 * q) = t;
        }
        __syncthreads();
    }
}

// COMPACT = 1: D has one row per INDEX (row j = the gradient of output row idx[j]) -- plain stores, no zero fill of an
// N x C matrix, no atomics; pad columns [C, cpad) of each row are written as zeros
template <int COMPACT>
__global__ __launch_bounds__(TPB) void ce_bwd_db_kernel(int C, const float* __restrict__ P, int64_t ldp,
                                                        const int* __restrict__ idx, int64_t n_idx,
                                                        const int* __restrict__ y, float inv_n,
                                                        float* __restrict__ D, int64_t ldd, float* __restrict__ part,
                                                        int cpad) {
    __shared__ float red[kWavesPerBlock][kWave];
    const int wv = threadIdx.x / kWave, lane = threadIdx.x % kWave;
    const int w = blockIdx.x * kWavesPerBlock + wv;
    float acc[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = 0.f;
    for (int64_t j = w; j < n_idx; j += kCeWaves) {
        const int64_t row = idx[j];
        const int yy = y[j];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int c = lane + kWave * k;
            if (c < C) {
                const float g = (P[row * ldp + c] - (c == yy ? 1.0f : 0.0f)) * inv_n;
                if constexpr (COMPACT) D[j * ldd + c] = g;
                else unsafeAtomicAdd(D + row * ldd + c, g);     // (adds onto zero / k identical values: exact, order-free)
                acc[k] += g;
            } else if (COMPACT && c < cpad) {
                D[j * ldd + c] = 0.f;
            }
        }
    }
    for (int k = 0; k < 16; ++k) {
        const int c = lane + kWave * k;
        if (kWave * k >= cpad) break;
        red[wv][lane] = acc[k];
        __syncthreads();
        if (wv == 0 && c < cpad) {
            float t = red[0][lane];
#pragma unroll
            for (int i = 1; i < kWavesPerBlock; ++i) t += red[i][lane];
            part[(int64_t)blockIdx.x * cpad + c] = (c < C) ? t : 0.f;
        }
        __syncthreads();
    }
}

// lasagne.updates.adam: a_t = lr * sqrt(1 - b2^t) / (1 - b1^t), computed in fp32 like floatX=float32
__host__ __device__ __forceinline__ float adam_a_t(float lr, float b1, float b2, float t) {
    return lr * sqrtf(1.0f - powf(b2, t)) / (1.0f - powf(b1, t));
}
// captured steps: t lives on the device.  state[0] = step count (incremented here), state[1] = bits of a_t
__global__ void adam_next_step_kernel(int64_t* state, float lr, float b1, float b2) {
    const int64_t t = state[0] + 1;
    state[0] = t;
    reinterpret_cast<float*>(state + 1)[0] = adam_a_t(lr, b1, b2, (float)t);
}

__global__ __launch_bounds__(TPB) void adam_kernel(int64_t n, float* __restrict__ p, float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v,
                                                   const float* __restrict__ regmask, float a_t,
                                                   const float* __restrict__ a_t_dev, float b1, float b2,
                                                   float eps, float l1, float l2) {
    if (a_t_dev) a_t = *a_t_dev;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) {
        const float pi = p[i];
        float gi = g[i];
        if (regmask) {
            const float sgn = (pi > 0.f) ? 1.f : ((pi < 0.f) ? -1.f : 0.f);
            gi += regmask[i] * (l1 * sgn + 2.0f * l2 * pi);
            g[i] = gi;              // g now holds d(train_loss)/dp including the penalty
        }
        const float mi = b1 * m[i] + (1.0f - b1) * gi;
        const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] = pi - a_t * mi / (sqrtf(vi) + eps);
    }
}

__global__ __launch_bounds__(TPB) void reg_partial_kernel(int64_t n, const float* __restrict__ p,
                                                          const float* __restrict__ regmask, float l1, float l2,
                                                          float* __restrict__ part) {
    __shared__ float s[kWavesPerBlock];
    float a = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) {
        const float pi = p[i];
        const float r = regmask ? regmask[i] : 1.f;
        a += r * (l1 * fabsf(pi) + l2 * pi * pi);
    }
    a = wave_sum(a);
    if (threadIdx.x % kWave == 0) s[threadIdx.x / kWave] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int i = 0; i < kWavesPerBlock; ++i) t += s[i];
        part[blockIdx.x] = t;
    }
}
__global__ void scalar_final_kernel(int nparts, const float* __restrict__ part, float* __restrict__ out) {
    float a = 0.f;                       // (one wave, fixed association: see pair_final_kernel)
    for (int i = threadIdx.x; i < nparts; i += kWave) a += part[i];
    a = wave_sum(a);
    if (threadIdx.x == 0) out[0] = a;
}

constexpr int kRegParts = 256;

}  // namespace

int softmax_rows_indexed_launch(const int* rows_dev, int n_list, int C, float* P, int64_t ldp, int* amax, hipStream_t st) {
    if (n_list <= 0 || C <= 0) return 0;
    const int ldp_pad = (int)std::min<int64_t>(ldp, (int64_t)((C + 3) / 4) * 4);
    GEOGCN_REQUIRE(ldp_pad <= 64 * 16, GEOGCN_E_ARG, "softmax of listed rows: C=%d > 1024", C);
    const dim3 grid((unsigned)cdiv(n_list, kWavesPerBlock));
    if (ldp_pad <= 64 * 4)
        hipLaunchKernelGGL((softmax_rows_reg_kernel<4>), grid, dim3(TPB), 0, st, (int64_t)n_list, C, (const float*)P, ldp, P, ldp, ldp_pad, amax, rows_dev);
    else
        hipLaunchKernelGGL((softmax_rows_reg_kernel<16>), grid, dim3(TPB), 0, st, (int64_t)n_list, C, (const float*)P, ldp, P, ldp, ldp_pad, amax, rows_dev);
    GEOGCN_LAUNCH_CHECK("softmax_rows_reg_kernel (listed rows)");
    return 0;
}
}  // namespace geogcn

using namespace geogcn;

extern "C" {

int geogcn_softmax_rows_f32(int64_t n, int32_t C, const float* logits, int64_t ldl, float* probs, int64_t ldp,
                            int32_t* argmax_out, void* stream) {
    GEOGCN_REQUIRE(n >= 0 && C >= 0, GEOGCN_E_SIZE, "softmax_rows_f32: negative size");
    if (n == 0 || C == 0) return 0;
    GEOGCN_REQUIRE(logits && probs, GEOGCN_E_NULL, "softmax_rows_f32: null pointer");
    GEOGCN_REQUIRE(ldl >= C && ldp >= C, GEOGCN_E_SIZE, "softmax_rows_f32: ld < C");
    const int ldp_pad = (int)std::min<int64_t>(ldp, (int64_t)((C + 3) / 4) * 4);
    const dim3 grid((unsigned)cdiv(n, kWavesPerBlock));
    hipStream_t st = (hipStream_t)stream;
    if (ldp_pad <= 64 * 4)
        hipLaunchKernelGGL((softmax_rows_reg_kernel<4>), grid, dim3(TPB), 0, st, n, C, logits, ldl, probs, ldp, ldp_pad, argmax_out);
    else if (ldp_pad <= 64 * 16)
        hipLaunchKernelGGL((softmax_rows_reg_kernel<16>), grid, dim3(TPB), 0, st, n, C, logits, ldl, probs, ldp, ldp_pad, argmax_out);
    else
        hipLaunchKernelGGL(softmax_rows_kernel, grid, dim3(TPB), 0, st, n, C, logits, ldl, probs, ldp, ldp_pad, argmax_out);
    GEOGCN_LAUNCH_CHECK("softmax_rows_kernel");
    return 0;
}

size_t geogcn_ce_metrics_workspace_bytes(int64_t n_idx) {
    if (n_idx <= 0) return 0;
    return (size_t)cdiv(n_idx, TPB) * 2 * sizeof(float);
}

int geogcn_ce_metrics_f32(int32_t C, const float* probs, int64_t ldp, const int32_t* argmax, const int32_t* idx,
                          int64_t n_idx, const int32_t* y, float* out2, void* ws, size_t ws_bytes, void* stream) {
    GEOGCN_REQUIRE(C > 0 && n_idx >= 0, GEOGCN_E_SIZE, "ce_metrics_f32: bad size");
    GEOGCN_REQUIRE(out2, GEOGCN_E_NULL, "ce_metrics_f32: null out");
    hipStream_t st = (hipStream_t)stream;
    if (n_idx == 0) {
        { const int zrc = zero_fill_async(out2, 2 * sizeof(float), st); if (zrc) return zrc; }
        return 0;
    }
    GEOGCN_REQUIRE(probs && idx && y, GEOGCN_E_NULL, "ce_metrics_f32: null pointer");
    const int nparts = (int)cdiv(n_idx, TPB);
    GEOGCN_REQUIRE(ws && ws_bytes >= (size_t)nparts * 2 * sizeof(float), GEOGCN_E_ARG,
                   "ce_metrics_f32: workspace too small");
    hipLaunchKernelGGL(ce_partial_kernel, dim3((unsigned)nparts), dim3(TPB), 0, st, C, probs, ldp, argmax, idx, n_idx,
                       y, (float*)ws);
    GEOGCN_LAUNCH_CHECK("ce_partial_kernel");
    hipLaunchKernelGGL(pair_final_kernel, dim3(1), dim3(kWave), 0, st, nparts, (const float*)ws, out2);
    GEOGCN_LAUNCH_CHECK("pair_final_kernel");
    return 0;
}

int geogcn_softmax_ce_bwd_f32(int64_t n, int32_t C, const float* probs, int64_t ldp, const int32_t* idx,
                              int64_t n_idx, const int32_t* y, float inv_n, float* dlogits, int64_t ldd,
                              void* stream) {
    GEOGCN_REQUIRE(n >= 0 && C >= 0 && n_idx >= 0, GEOGCN_E_SIZE, "softmax_ce_bwd_f32: negative size");
    if (n == 0 || C == 0) return 0;
    GEOGCN_REQUIRE(dlogits, GEOGCN_E_NULL, "softmax_ce_bwd_f32: null dlogits");
    GEOGCN_REQUIRE(ldd >= C, GEOGCN_E_SIZE, "softmax_ce_bwd_f32: ldd < C");
    hipStream_t st = (hipStream_t)stream;
    { const int zrc = zero_fill_async(dlogits, (size_t)n * (size_t)ldd * sizeof(float), st); if (zrc) return zrc; }
    if (n_idx == 0) return 0;
    GEOGCN_REQUIRE(probs && idx && y, GEOGCN_E_NULL, "softmax_ce_bwd_f32: null pointer");
    hipLaunchKernelGGL(ce_bwd_kernel, dim3((unsigned)cdiv(n_idx, kWavesPerBlock)), dim3(TPB), 0, st, C, probs, ldp, idx,
                       n_idx, y, inv_n, dlogits, ldd);
    GEOGCN_LAUNCH_CHECK("ce_bwd_kernel");
    return 0;
}

size_t geogcn_softmax_ce_bwd_db_workspace_bytes(int32_t C) {
    return C <= 0 ? 0 : (size_t)kCeBlocks * (size_t)((C + 3) / 4) * 4 * sizeof(float);
}

int geogcn_softmax_ce_bwd_db_f32(int64_t n, int32_t C, const float* probs, int64_t ldp, const int32_t* idx,
                                 int64_t n_idx, const int32_t* y, float inv_n, float* dlogits, int64_t ldd, float* db,
                                 void* ws, size_t ws_bytes, void* stream) {
    GEOGCN_REQUIRE(n >= 0 && C >= 0 && n_idx >= 0, GEOGCN_E_SIZE, "softmax_ce_bwd_db_f32: negative size");
    if (C == 0) return 0;
    if (n == 0) {
        // a rank that owns no rows: its share of the bias gradient is the zero vector, not last step's (already
        // all-reduced) value -- found by the random models on 4 and 8 ranks with 5 nodes
        return db ? zero_fill_async(db, (size_t)((C + 3) / 4 * 4) * sizeof(float), (hipStream_t)stream) : 0;
    }
    GEOGCN_REQUIRE(dlogits && db, GEOGCN_E_NULL, "softmax_ce_bwd_db_f32: null pointer");
    GEOGCN_REQUIRE(ldd >= C, GEOGCN_E_SIZE, "softmax_ce_bwd_db_f32: ldd < C");
    GEOGCN_REQUIRE(C <= 16 * kWave, GEOGCN_E_ARG, "softmax_ce_bwd_db_f32: C=%d > %d", C, 16 * kWave);
    hipStream_t st = (hipStream_t)stream;
    { const int zrc = zero_fill_async(dlogits, (size_t)n * (size_t)ldd * sizeof(float), st); if (zrc) return zrc; }
    const int cpad = (C + 3) / 4 * 4;
    if (n_idx == 0) return zero_fill_async(db, (size_t)cpad * sizeof(float), st);
    GEOGCN_REQUIRE(probs && idx && y, GEOGCN_E_NULL, "softmax_ce_bwd_db_f32: null pointer");
    GEOGCN_REQUIRE(ws && ws_bytes >= geogcn_softmax_ce_bwd_db_workspace_bytes(C), GEOGCN_E_ARG,
                   "softmax_ce_bwd_db_f32: workspace too small");
    hipLaunchKernelGGL(ce_bwd_db_kernel<0>, dim3(kCeBlocks), dim3(TPB), 0, st, C, probs, ldp, idx, n_idx, y,
                       inv_n, dlogits, ldd, (float*)ws, cpad);
    GEOGCN_LAUNCH_CHECK("ce_bwd_db_kernel");
    return colsum_final_launch(kCeBlocks, C, (const float*)ws, cpad, db, st);
}

int geogcn_softmax_ce_rows_bwd_db_f32(int32_t C, const float* probs, int64_t ldp, const int32_t* idx, int64_t n_idx,
                                      const int32_t* y, float inv_n, float* drows, int64_t ldd, float* db, void* ws,
                                      size_t ws_bytes, void* stream) {
    const char* fn = "softmax_ce_rows_bwd_db_f32";
    GEOGCN_REQUIRE(C >= 0 && n_idx >= 0, GEOGCN_E_SIZE, "%s: negative size", fn);
    if (C == 0) return 0;
    GEOGCN_REQUIRE(db, GEOGCN_E_NULL, "%s: null db", fn);
    const int cpad = (C + 3) / 4 * 4;
    hipStream_t st = (hipStream_t)stream;
    if (n_idx == 0) return zero_fill_async(db, (size_t)cpad * sizeof(float), st);
    GEOGCN_REQUIRE(probs && idx && y && drows, GEOGCN_E_NULL, "%s: null pointer", fn);
    GEOGCN_REQUIRE(ldd >= cpad, GEOGCN_E_SIZE, "%s: ldd < roundup4(C)", fn);
    GEOGCN_REQUIRE(C <= 16 * kWave, GEOGCN_E_ARG, "%s: C=%d > %d", fn, C, 16 * kWave);
    GEOGCN_REQUIRE(ws && ws_bytes >= geogcn_softmax_ce_bwd_db_workspace_bytes(C), GEOGCN_E_ARG, "%s: workspace too small", fn);
    if (C % 4 == 0 && ldp % 4 == 0 && ldd % 4 == 0 && aligned16(probs) && aligned16(drows) && aligned16(ws)) {
        hipLaunchKernelGGL(ce_rows4_db_kernel, dim3(kCeBlocks), dim3(TPB), 0, st, C / 4, probs, ldp, idx, n_idx, y, inv_n, drows,
                           ldd, (float*)ws);
        GEOGCN_LAUNCH_CHECK("ce_rows4_db_kernel");
        return colsum_final_launch(kCeBlocks, C, (const float*)ws, cpad, db, st);
    }
    hipLaunchKernelGGL(ce_bwd_db_kernel<1>, dim3(kCeBlocks), dim3(TPB), 0, st, C, probs, ldp, idx, n_idx, y, inv_n, drows, ldd,
                       (float*)ws, cpad);
    GEOGCN_LAUNCH_CHECK("ce_bwd_db_kernel");
    return colsum_final_launch(kCeBlocks, C, (const float*)ws, cpad, db, st);
}

int geogcn_adam_step_f32(int64_t n, float* p, float* g, float* m, float* v, const float* regmask, float lr,
                         float b1, float b2, float eps, int32_t t, float l1, float l2, void* stream) {
    GEOGCN_REQUIRE(n >= 0 && t >= 1, GEOGCN_E_SIZE, "adam_step_f32: n=%lld t=%d", (long long)n, t);
    if (n == 0) return 0;
    GEOGCN_REQUIRE(p && g && m && v, GEOGCN_E_NULL, "adam_step_f32: null pointer");
    const float a_t = adam_a_t(lr, b1, b2, (float)t);
    const int64_t blocks = std::min<int64_t>(cdiv(n, TPB), (int64_t)kNumCU * 8);
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(TPB), 0, (hipStream_t)stream, n, p, g, m, v,
                       (l1 != 0.f || l2 != 0.f) ? regmask : nullptr, a_t, (const float*)nullptr, b1, b2, eps, l1, l2);
    GEOGCN_LAUNCH_CHECK("adam_kernel");
    return 0;
}

int geogcn_adam_step_ctr_f32(int64_t n, float* p, float* g, float* m, float* v, const float* regmask, float lr,
                             float b1, float b2, float eps, int64_t* state_dev, float l1, float l2, void* stream) {
    GEOGCN_REQUIRE(n >= 0, GEOGCN_E_SIZE, "adam_step_ctr_f32: n=%lld", (long long)n);
    GEOGCN_REQUIRE(state_dev, GEOGCN_E_NULL, "adam_step_ctr_f32: null state");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(adam_next_step_kernel, dim3(1), dim3(1), 0, st, state_dev, lr, b1, b2);
    GEOGCN_LAUNCH_CHECK("adam_next_step_kernel");
    if (n == 0) return 0;
    GEOGCN_REQUIRE(p && g && m && v, GEOGCN_E_NULL, "adam_step_ctr_f32: null pointer");
    const int64_t blocks = std::min<int64_t>(cdiv(n, TPB), (int64_t)kNumCU * 8);
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(TPB), 0, st, n, p, g, m, v,
                       (l1 != 0.f || l2 != 0.f) ? regmask : nullptr, 0.f, reinterpret_cast<const float*>(state_dev + 1), b1,
                       b2, eps, l1, l2);
    GEOGCN_LAUNCH_CHECK("adam_kernel");
    return 0;
}

int geogcn_reg_penalty_f32(int64_t n, const float* p, const float* regmask, float l1, float l2, float* out, void* ws,
                           size_t ws_bytes, void* stream) {
    GEOGCN_REQUIRE(n >= 0, GEOGCN_E_SIZE, "reg_penalty_f32: negative size");
    GEOGCN_REQUIRE(out, GEOGCN_E_NULL, "reg_penalty_f32: null out");
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) {
        { const int zrc = zero_fill_async(out, sizeof(float), st); if (zrc) return zrc; }
        return 0;
    }
    GEOGCN_REQUIRE(p, GEOGCN_E_NULL, "reg_penalty_f32: null p");
    GEOGCN_REQUIRE(ws && ws_bytes >= kRegParts * sizeof(float), GEOGCN_E_ARG, "reg_penalty_f32: workspace < %zu bytes",
                   kRegParts * sizeof(float));
    const int nparts = (int)std::min<int64_t>(kRegParts, cdiv(n, TPB));
    hipLaunchKernelGGL(reg_partial_kernel, dim3((unsigned)nparts), dim3(TPB), 0, st, n, p, regmask, l1, l2, (float*)ws);
    GEOGCN_LAUNCH_CHECK("reg_partial_kernel");
    hipLaunchKernelGGL(scalar_final_kernel, dim3(1), dim3(kWave), 0, st, nparts, (const float*)ws, out);
    GEOGCN_LAUNCH_CHECK("scalar_final_kernel");
    return 0;
}

}  // extern "C"
