// CSR x dense SpMM for gfx950 -- replaces Theano's StructuredDot C loop on the GCN hot path
// (reference gcnmodel.py:39,130,153 and the StructuredDot gradients autodiff derives).
//
// HBM-bound gather kernel (AI ~5.6 flop/B): no MFMA, no reshaping into GEMM.
//   * a GROUP of 16 lanes owns one CSR row; each lane keeps K4 float4 accumulators, so one
//     nonzero turns into K4 fully coalesced 256-byte reads of the gathered B row per group and a
//     64-lane wave keeps 4 independent rows (gather streams) in flight;
//   * (col, val) pairs are read 16 at a time, coalesced, and handed round the group with
//     width-16 shuffles (ds_bpermute: the LDS crossbar is otherwise idle here);
//   * accumulation is sequential in stored index order with fmaf -- the same order as the
//     reference's row loop -- so short rows are bitwise reproducible run to run;
//   * rows longer than plan->long_row_nnz are cut into fixed chunks, each chunk is a "virtual
//     row" writing a partial sum to the workspace, and a third kernel adds the partials in chunk
//     order (no float atomics anywhere);
//   * bias + tanh/sigmoid epilogue fused into the store (gcnmodel.py:41-42,132-136).
#include "common.h"

#include <stdlib.h>

#include <algorithm>
#include <vector>

namespace geogcn {
namespace {

constexpr int kGroup = 16;                // lanes per row (wide operands); narrow ones use 8
constexpr int kBlock = 256;               // 4 waves = 16 groups
// threads per row-kernel workgroup: 512 (32 rows of 16 lanes) measured 1.7 % faster than 256, 128 slower
// (pinned graph 1.851 / 1.883 / 1.890 ms, community graph 1.080 / 1.100 / 1.118); the long-row combine keeps kBlock
constexpr int kRowBlock = 512;
constexpr int kRowAlign = kRowBlock / 8;     // rows per workgroup of the narrowest (8-lane) variant: XCD ranges are multiples of it


struct F4 {
    float x, y, z, w;
};

__device__ __forceinline__ void fma4(float4& acc, float a, const float4& b) {
    acc.x = fmaf(a, b.x, acc.x);
    acc.y = fmaf(a, b.y, acc.y);
    acc.z = fmaf(a, b.z, acc.z);
    acc.w = fmaf(a, b.w, acc.w);
}

typedef float f32x4v __attribute__((ext_vector_type(4)));

// (a hub hint -- hub columns first with default loads, the rest with non-temporal loads so that they do not evict the
//  hubs -- was built and measured in rounds 1-2: 2.0 -> 2.6-2.8 ms, removed in round 3; DESIGN_NOTEBOOK.md section 4.1)
__device__ __forceinline__ float4 gather4(const void* row, int q) {
    const f32x4v v = *(reinterpret_cast<const f32x4v*>(row) + q);
    return make_float4(v.x, v.y, v.z, v.w);
}
// BF = 1: the gathered operand is stored as bfloat16.  A lane still loads 16 bytes per pass -- now 8
// features, unpacked into TWO float4 accumulators -- because this kernel's ceiling counts 64-lane load
// instructions in flight, not bytes: a first version with 8-byte loads (4 features per lane and pass) moved
// half the bytes per instruction and ran no faster than fp32 (1.93 vs 1.98 ms at F = 300).
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4v gather8_bf16(const void* row, int q8) {
    return *(reinterpret_cast<const u32x4v*>(row) + q8);
}
// a * (8 bf16 features) accumulated into two float4s; the raw 16 bytes stay packed until here, so that the
// loads in flight cost 4 VGPRs each, not 8
__device__ __forceinline__ void fma8_bf16(float4& lo, float4& hi, float a, const u32x4v v) {
    fma4(lo, a, make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u),
                            __uint_as_float(v.y << 16), __uint_as_float(v.y & 0xffff0000u)));
    fma4(hi, a, make_float4(__uint_as_float(v.z << 16), __uint_as_float(v.z & 0xffff0000u),
                            __uint_as_float(v.w << 16), __uint_as_float(v.w & 0xffff0000u)));
}
// float4 index (within the output row) of accumulator k of lane `lane`: fp32 -> pass k covers float4s
// lane + G*k; bf16 -> accumulators 2j, 2j+1 are the two halves of the 8 features of pass j
template <int G, int BF>
__device__ __forceinline__ int f4_index(int lane, int k) {
    if constexpr (BF) return 2 * (lane + G * (k >> 1)) + (k & 1);
    else return lane + G * k;
}
template <int BF>
__device__ __forceinline__ const void* row_ptr(const void* B, int64_t ldb, int c) {
    if constexpr (BF) return reinterpret_cast<const uint16_t*>(B) + (int64_t)c * ldb;
    else return reinterpret_cast<const float*>(B) + (int64_t)c * ldb;
}

// One group walks nonzeros [s, e) and accumulates into acc[K4].
template <int K4, int G, int BF>
__device__ __forceinline__ void group_accumulate(int s, int e, int lane16, int nF4,
                                                 const int* __restrict__ colidx,
                                                 const float* __restrict__ val,
                                                 const void* __restrict__ B, int64_t ldb,
                                                 float4 (&acc)[K4]) {
    // (col, val) of batch i+1 are loaded before batch i's gathers are issued: a row of the TwitterUS-shape graph has 24
    // entries on average = two batches, and the second index load would otherwise sit between two gather phases
    int c = 0;
    float a = 0.f;
    if (s + lane16 < e) {
        c = colidx[s + lane16];
        a = val[s + lane16];
    }
    for (int base = s; base < e; base += G) {
        const int jn = base + G + lane16;
        int cn = 0;
        float an = 0.f;
        if (jn < e) {
            cn = colidx[jn];
            an = val[jn];
        }
        const int cnt = min(G, e - base);
        int t = 0;
        // two nonzeros per trip: 2*K4 independent loads in flight per lane
        for (; t + 1 < cnt; t += 2) {
            const int c0 = __shfl(c, t, G);
            const int c1 = __shfl(c, t + 1, G);
            const float a0 = __shfl(a, t, G);
            const float a1 = __shfl(a, t + 1, G);
            const void* b0 = row_ptr<BF>(B, ldb, c0);
            const void* b1 = row_ptr<BF>(B, ldb, c1);
            if constexpr (BF) {
                u32x4v r0[K4 / 2], r1[K4 / 2];
#pragma unroll
                for (int j = 0; j < K4 / 2; ++j) {
                    const int q8 = lane16 + G * j;
                    if (2 * q8 < nF4) {
                        r0[j] = gather8_bf16(b0, q8);
                        r1[j] = gather8_bf16(b1, q8);
                    }
                }
#pragma unroll
                for (int j = 0; j < K4 / 2; ++j) {
                    // (the upper half of the last piece may lie in the zero pad columns of the operand)
                    if (2 * (lane16 + G * j) < nF4) {
                        fma8_bf16(acc[2 * j], acc[2 * j + 1], a0, r0[j]);
                        fma8_bf16(acc[2 * j], acc[2 * j + 1], a1, r1[j]);
                    }
                }
            } else {
                float4 v0[K4], v1[K4];
#pragma unroll
                for (int k = 0; k < K4; ++k) {
                    const int q = lane16 + G * k;
                    if (q < nF4) {
                        v0[k] = gather4(b0, q);
                        v1[k] = gather4(b1, q);
                    }
                }
#pragma unroll
                for (int k = 0; k < K4; ++k) {
                    const int q = lane16 + G * k;
                    if (q < nF4) {
                        fma4(acc[k], a0, v0[k]);
                        fma4(acc[k], a1, v1[k]);
                    }
                }
            }
        }
        if (t < cnt) {
            const int c0 = __shfl(c, t, G);
            const float a0 = __shfl(a, t, G);
            const void* b0 = row_ptr<BF>(B, ldb, c0);
            if constexpr (BF) {
#pragma unroll
                for (int k = 0; k < K4; k += 2) {
                    const int q8 = lane16 + G * (k >> 1);
                    if (2 * q8 < nF4) fma8_bf16(acc[k], acc[k + 1], a0, gather8_bf16(b0, q8));
                }
            } else {
#pragma unroll
                for (int k = 0; k < K4; ++k) {
                    const int q = lane16 + G * k;
                    if (q < nF4) fma4(acc[k], a0, gather4(b0, q));
                }
            }
        }
        c = cn;
        a = an;
    }
}

template <int ACT>
__device__ __forceinline__ float4 epilogue4(float4 r, int col0, int F, const float* __restrict__ bias, bool bias_vec) {
    float o[4] = {r.x, r.y, r.z, r.w};
    const float4 b4 = load_bias4(bias, col0, F, bias_vec);
    const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int col = col0 + i;
        if (col < F) {
            float x = o[i];
            if (bias) x += bb[i];
            o[i] = apply_act<ACT>(x);
        } else {
            o[i] = 0.f;             // keep pad columns zero (geogcn.h convention)
        }
    }
    return make_float4(o[0], o[1], o[2], o[3]);
}

// Highway epilogue (gcnmodel.py:266 fused into the producing SpMM): with hw.T != nullptr the kernel stores
// Hc = act(A.B + bias) to C as usual AND Hout = T*Hc + (1-T)*H, reading the T and H rows it needs (coalesced,
// row-contiguous) instead of a separate elementwise pass re-reading Hc.
struct HwArgs {
    const float* T;
    const float* H;
    float* Hout;
    int64_t ld;          // common pitch of T, H, Hout
    const float* init;   // nullable: the accumulators start from init[row][:] (pitch ld_init) instead of zero --
    int64_t ld_init;     // C = act(C0 + A.B + bias): a product that continues one already in C (X.W0: dense head + tail)
    int* amax = nullptr; // softmax epilogue (HW = 2, geogcn_spmm_csr_softmax_f32): C receives softmax(A.B + bias) row by row, amax
    int softmax = 0;     // (nullable) the first index of each row's maximum
};
__device__ __forceinline__ float4 highway_mix(const float4 t, const float4 hc, const float4 h) {
    return make_float4(t.x * hc.x + (1.0f - t.x) * h.x, t.y * hc.y + (1.0f - t.y) * h.y,
                       t.z * hc.z + (1.0f - t.z) * h.z, t.w * hc.w + (1.0f - t.w) * h.w);
}

struct XcdRows {
    int lo[kNumXCD + 1];         // XCD x owns rows [lo[x], lo[x + 1])
};
constexpr int kIdleSlot = INT32_MIN;

// ONE launch covers every stored edge.  Block b runs on XCD b % 8 (measured; speed only) and is the (b / 8)-th block of
// that XCD's SCHEDULE: XCD x takes the CONTIGUOUS row range [lo[x], lo[x+1]) -- one CSR row per 16-lane group, fused
// epilogue, rows longer than long_row_nnz skipped -- and the 128-nonzero chunks of the long rows that lie in that range
// (raw partial sums into the workspace P, combined by spmm_long_reduce_kernel), a chunk block taking its turn where the
// long row itself sits in the sweep.  When the node numbering has locality (geographconv_amd.graph: community
// reordering) the rows an XCD works on at any moment share their neighbours and the gathered rows of B stay in that
// XCD's 4 MB L2 -- the hub rows' chunks included (until round 3 they ran in the leading blocks of the launch on arbitrary
// XCDs, outside the window: 13 % of the entries of the TwitterUS-shape graphs).  The ranges hold equal numbers of stored
// entries, not of rows (a hub-first numbering would otherwise give one XCD all the work).
// `sched` (host-built per plan, nullable when there are no long rows): entry >= 0 = row block v of the XCD's range,
// < 0 = chunk block -1 - v (chunks [c * rows-per-block, ...) in row order), kIdleSlot = nothing.  Which XCD walks which chunk
// block, and when, is the host's choice (geogcn_spmm_plan_create).  per_xcd = 0 and no schedule: plain b -> row block map.
template <int K4, int ACT, int G, int BF, int HW = 0>
__global__ __launch_bounds__(kRowBlock) void spmm_rows_kernel(
    int n_rows, const int* __restrict__ rowptr, const int* __restrict__ colidx,
    const float* __restrict__ val, const void* __restrict__ B, int64_t ldb, float* __restrict__ C,
    int64_t ldc, int F, const float* __restrict__ bias, int long_row_nnz, const int* __restrict__ sched, int n_chunks,
    const int* __restrict__ chunk_start, const int* __restrict__ chunk_end, float* __restrict__ P, int64_t ldp,
    const HwArgs hw, const int per_xcd, const XcdRows xr) {
    constexpr int kGpb = kRowBlock / G;       // groups (= rows, = chunks) per block
    const int lane16 = threadIdx.x % G;
    const int nF4 = (F + 3) >> 2;
    const bool bias_vec = (reinterpret_cast<uintptr_t>(bias) & 15u) == 0;
    float4 acc[K4];
#pragma unroll
    for (int k = 0; k < K4; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int x = blockIdx.x % kNumXCD;
    int v = blockIdx.x / kNumXCD;             // without a schedule: the v-th row block of XCD x
    if (sched) {
        v = sched[blockIdx.x];
        if (v == kIdleSlot) return;
    }
    if (v < 0) {
        const int ch = (-1 - v) * kGpb + (threadIdx.x / G);
        if (ch >= n_chunks) return;
        const int cs = chunk_start[ch], ce = chunk_end[ch];
        group_accumulate<K4, G, BF>(cs, ce, lane16, nF4, colidx, val, B, ldb, acc);
        float4* out = reinterpret_cast<float4*>(P + (int64_t)ch * ldp);
#pragma unroll
        for (int k = 0; k < K4; ++k) {
            const int q = f4_index<G, BF>(lane16, k);
            if (q < nF4) out[q] = acc[k];
        }
        return;
    }
    int row;
    if (per_xcd > 0) {
        // (lo[x] is a multiple of kRowAlign or n_rows itself: counting from lo[x] a block never reaches below its range --
        //  harmless for a plain product, which would just be written twice, but the accumulate form reads what it writes)
        row = xr.lo[x] + v * kGpb + (threadIdx.x / G);
        if (row >= xr.lo[x + 1]) return;
    } else {
        row = (int)blockIdx.x * kGpb + (threadIdx.x / G);
        if (row >= n_rows) return;
    }
    const int s = rowptr[row];
    const int e = rowptr[row + 1];
    if (e - s > long_row_nnz) return;       // a long row: its chunks are separate entries of the schedule
    if (hw.init) {
        const float4* irow = reinterpret_cast<const float4*>(hw.init + (int64_t)row * hw.ld_init);
#pragma unroll
        for (int k = 0; k < K4; ++k) {
            const int q = f4_index<G, BF>(lane16, k);
            if (q < nF4) acc[k] = irow[q];
        }
    }
    group_accumulate<K4, G, BF>(s, e, lane16, nF4, colidx, val, B, ldb, acc);
    float4* out = reinterpret_cast<float4*>(C + (int64_t)row * ldc);
    if constexpr (HW == 2) {
        // softmax of the row in the epilogue (the output layer: gcnmodel.py:149 + nonlinearity softmax): the row lives in the
        // G lanes of its group, 4 K4 values each -- maximum and first index attaining it, exp(x - max), sum, quotient: the
        // arithmetic of softmax_rows_reg_kernel (softmax_adam.hip) with the sum taken over this layout's partial sums
        float v[K4][4];
        float m = -INFINITY;
        int mi = 0x7fffffff;
#pragma unroll
        for (int k = 0; k < K4; ++k) {
            const int q = f4_index<G, BF>(lane16, k);
            const float a4[4] = {acc[k].x, acc[k].y, acc[k].z, acc[k].w};
            const float4 b4 = q < nF4 ? load_bias4(bias, q * 4, F, bias_vec) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int col = q * 4 + i;
                float x = -INFINITY;
                if (q < nF4 && col < F) {
                    x = a4[i];
                    if (bias) x += bb[i];
                }
                v[k][i] = x;
                if (x > m) { m = x; mi = col; }          // ascending columns within a lane: strict > keeps the first index
            }
        }
#pragma unroll
        for (int o = G / 2; o > 0; o >>= 1) {
            const float om = __shfl_xor(m, o, G);
            const int oi = __shfl_xor(mi, o, G);
            if (om > m || (om == m && oi < mi)) { m = om; mi = oi; }
        }
        float ssum = 0.f;
#pragma unroll
        for (int k = 0; k < K4; ++k)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v[k][i] = (v[k][i] == -INFINITY) ? 0.f : expf(v[k][i] - m);
                ssum += v[k][i];
            }
#pragma unroll
        for (int o = G / 2; o > 0; o >>= 1) ssum += __shfl_xor(ssum, o, G);
#pragma unroll
        for (int k = 0; k < K4; ++k) {
            const int q = f4_index<G, BF>(lane16, k);
            if (q < nF4) out[q] = make_float4(v[k][0] / ssum, v[k][1] / ssum, v[k][2] / ssum, v[k][3] / ssum);
        }
        if (hw.amax && lane16 == 0) hw.amax[row] = mi;
        return;
    }
#pragma unroll
    for (int k = 0; k < K4; ++k) {
        const int q = f4_index<G, BF>(lane16, k);
        if (q < nF4) {
            const float4 hc = epilogue4<ACT>(acc[k], q * 4, F, bias, bias_vec);
            out[q] = hc;
            if constexpr (HW == 1) {
#ifdef GEOGCN_SPMM_HW_NT          // A/B build only: streaming cache policy for the epilogue's operands (1: T / H loads, 2: Hout store)
                typedef float f4v_ __attribute__((ext_vector_type(4)));
                const f4v_* tp = reinterpret_cast<const f4v_*>(hw.T + (int64_t)row * hw.ld) + q;
                const f4v_* hp = reinterpret_cast<const f4v_*>(hw.H + (int64_t)row * hw.ld) + q;
                const f4v_ tv = (GEOGCN_SPMM_HW_NT & 1) ? __builtin_nontemporal_load(tp) : *tp;
                const f4v_ hv = (GEOGCN_SPMM_HW_NT & 1) ? __builtin_nontemporal_load(hp) : *hp;
                const float4 mix = highway_mix(make_float4(tv.x, tv.y, tv.z, tv.w), hc, make_float4(hv.x, hv.y, hv.z, hv.w));
                f4v_* op = reinterpret_cast<f4v_*>(hw.Hout + (int64_t)row * hw.ld) + q;
                const f4v_ mv = {mix.x, mix.y, mix.z, mix.w};
                if (GEOGCN_SPMM_HW_NT & 2) __builtin_nontemporal_store(mv, op);
                else *op = mv;
#else
                const float4 t = reinterpret_cast<const float4*>(hw.T + (int64_t)row * hw.ld)[q];
                const float4 h = reinterpret_cast<const float4*>(hw.H + (int64_t)row * hw.ld)[q];
                reinterpret_cast<float4*>(hw.Hout + (int64_t)row * hw.ld)[q] = highway_mix(t, hc, h);
#endif
            }
        }
    }
}

// long rows: one block per row.  The hub of the TwitterUS-shape graph has 587 partial rows: one thread per column
// adding them one after the other was latency-bound (0.15 ms for 28 MB of partials).  Four wave-wide chunk
// groups (one block per row and 64-column slice) walk the partials in an interleaved, FIXED order (group g takes chunks
// g, g+4, ...; four loads in flight each) and the four sums are combined as (s0+s1)+(s2+s3): no atomics, the
// same association every run.
template <int ACT, int HW = 0>
__global__ __launch_bounds__(kBlock) void spmm_long_reduce_kernel(
    const int* __restrict__ long_rows, const int* __restrict__ long_first, const float* __restrict__ P,
    int64_t ldp, float* __restrict__ C, int64_t ldc, int F, int Fpad, const float* __restrict__ bias,
    const HwArgs hw) {
    __shared__ float red[4][kWave];
    const int lr = blockIdx.x;
    const int row = long_rows[lr];
    const int c0 = long_first[lr];
    const int c1 = long_first[lr + 1];
    const int cg = threadIdx.x / kWave, lane = threadIdx.x % kWave;
    {
        const int col = blockIdx.y * kWave + lane;          // one 64-column slice per block (grid.y slices)
        float acc = 0.f;
        if (col < Fpad) {
            int ch = c0 + cg;
            for (; ch + 12 < c1; ch += 16) {
                const float p0 = P[(int64_t)(ch + 0) * ldp + col];
                const float p1 = P[(int64_t)(ch + 4) * ldp + col];
                const float p2 = P[(int64_t)(ch + 8) * ldp + col];
                const float p3 = P[(int64_t)(ch + 12) * ldp + col];
                acc = (((acc + p0) + p1) + p2) + p3;
            }
            for (; ch < c1; ch += 4) acc += P[(int64_t)ch * ldp + col];
        }
        red[cg][lane] = acc;
        __syncthreads();
        if (cg == 0 && col < Fpad) {
            acc = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
            if (hw.init) acc += hw.init[(int64_t)row * hw.ld_init + col];
            float o = 0.f;
            if (col < F) {
                if (bias) acc += bias[col];
                o = apply_act<ACT>(acc);
            }
            C[(int64_t)row * ldc + col] = o;
            if constexpr (HW) {
                const float t = hw.T[(int64_t)row * hw.ld + col], h = hw.H[(int64_t)row * hw.ld + col];
                hw.Hout[(int64_t)row * hw.ld + col] = t * o + (1.0f - t) * h;
            }
        }
    }
}

// Fallback for pitches that are not float4-addressable: one wave per row, scalar lanes.
template <int ACT>
__global__ __launch_bounds__(kBlock) void spmm_scalar_kernel(
    int n_rows, const int* __restrict__ rowptr, const int* __restrict__ colidx,
    const float* __restrict__ val, const float* __restrict__ B, int64_t ldb, float* __restrict__ C,
    int64_t ldc, int F, const float* __restrict__ bias) {
    const int row = blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave;
    const int lane = threadIdx.x % kWave;
    if (row >= n_rows) return;
    const int s = rowptr[row], e = rowptr[row + 1];
    for (int col = lane; col < F; col += kWave) {
        float acc = 0.f;
        for (int j = s; j < e; ++j) acc = fmaf(val[j], B[(int64_t)colidx[j] * ldb + col], acc);
        if (bias) acc += bias[col];
        C[(int64_t)row * ldc + col] = apply_act<ACT>(acc);
    }
}

template <int K4, int G, int BF>
int launch_k4(const geogcn_spmm_plan* plan, int n_rows, const int* rowptr, const int* colidx,
              const float* val, const void* B, int64_t ldb, float* C, int64_t ldc, int F,
              const float* bias, int act, float* ws, hipStream_t st, int64_t nnz, const HwArgs& hw);

}  // namespace
}  // namespace geogcn

struct geogcn_timer;
struct geogcn_spmm_plan {
    int32_t n_rows = 0;
    int32_t long_row_nnz = 0;
    int32_t chunk_nnz = 0;
    int64_t n_long = 0;
    int64_t n_chunks = 0;
    int* d_long_rows = nullptr;    // [n_long]
    int* d_long_first = nullptr;   // [n_long + 1] first chunk of each long row
    int* d_chunk_start = nullptr;  // [n_chunks]
    int* d_chunk_end = nullptr;    // [n_chunks]
    int xcd_lo[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};      // per-XCD row ranges (multiples of 64 = kRowAlign), equal stored entries
    int* d_sched[2] = {nullptr, nullptr};             // per-XCD block schedules for 16- / 8-lane groups (only with long rows)
    int sched_slots[2] = {0, 0};                      // blocks per XCD in each
    // profiling aid (geogcn_spmm_plan_attach_timer): the caller's event pool, sampled by the products that run on THIS plan
    geogcn_timer* timer = nullptr;
    int32_t timer_F = 0;
};

struct geogcn_timer {
    std::vector<hipEvent_t> begin, end;
    int used = 0;
};

namespace geogcn {
namespace {

template <int K4, int G, int BF>
int launch_k4(const geogcn_spmm_plan* plan, int n_rows, const int* rowptr, const int* colidx,
              const float* val, const void* B, int64_t ldb, float* C, int64_t ldc, int F,
              const float* bias, int act, float* ws, hipStream_t st, int64_t nnz, const HwArgs& hw) {
    constexpr int kGroupsPerBlock = kRowBlock / G;
    const int long_nnz = plan ? plan->long_row_nnz : INT32_MAX;
    const int n_chunks = (plan && plan->n_long > 0) ? (int)plan->n_chunks : 0;
    constexpr int xcd_rows = 1;      // (0 = plain block -> row-block map: the A/B of DESIGN_NOTEBOOK.md section 4.1)
    const int64_t ldp = (int64_t)((F + 3) / 4) * 4;
    const int n_row_blocks = (int)cdiv(n_rows, kGroupsPerBlock);
    XcdRows xr{};
    int per_xcd = 0;
    const int* sched = nullptr;
    if (xcd_rows && plan) {
        for (int x = 0; x <= kNumXCD; ++x) xr.lo[x] = plan->xcd_lo[x];
        for (int x = 0; x < kNumXCD; ++x) per_xcd = std::max(per_xcd, (int)cdiv(xr.lo[x + 1] - xr.lo[x], kGroupsPerBlock));
        if (n_chunks > 0) {       // row blocks and chunk blocks interleaved per XCD (host-built at plan creation)
            sched = plan->d_sched[G == 8 ? 1 : 0];
            per_xcd = plan->sched_slots[G == 8 ? 1 : 0];
        }
    } else if (xcd_rows) {
        per_xcd = (int)cdiv(cdiv(n_row_blocks, kNumXCD) * kGroupsPerBlock, kRowAlign) * kRowAlign / kGroupsPerBlock;
        for (int x = 0; x <= kNumXCD; ++x) xr.lo[x] = (int)std::min<int64_t>(n_rows, (int64_t)x * per_xcd * kGroupsPerBlock);
    }
    const dim3 grid((unsigned)(per_xcd ? per_xcd * kNumXCD : n_row_blocks));
    // the timer rides on the PLAN handle the caller passes (no library-global state): sampled are the products on this plan
    // whose width matches, except the fused highway launches (they move other bytes: not the kernel bench.py prices)
    geogcn_timer* tm = plan ? plan->timer : nullptr;
    const bool timed = tm && (plan->timer_F == 0 || plan->timer_F == F) && tm->used < (int)tm->begin.size() && n_rows > 0 && !hw.T;
    if (timed) GEOGCN_HIP(hipEventRecord(tm->begin[tm->used], st));
#define GEOGCN_ROWS(ACT) GEOGCN_ROWS__(ACT, 0)
#define GEOGCN_ROWS__(ACT, HW_)                                                                  \
    hipLaunchKernelGGL((spmm_rows_kernel<K4, ACT, G, BF, HW_>), grid, dim3(kRowBlock), 0, st, n_rows, rowptr,   \
                       colidx, val, B, ldb, C, ldc, F, bias, long_nnz, sched, n_chunks,          \
                       n_chunks ? plan->d_chunk_start : nullptr, n_chunks ? plan->d_chunk_end : nullptr, ws, ldp, hw, per_xcd, xr)
    if (n_rows > 0) {
        if (hw.softmax) {
            if constexpr (BF == 0 && G == kGroup && K4 <= 8) { GEOGCN_ROWS__(GEOGCN_ACT_NONE, 2); }      // (checked by the caller)
        }
        else if (hw.T) { GEOGCN_ROWS__(GEOGCN_ACT_TANH, 1); }        // highway epilogue: tanh branch only (checked by the caller)
        else if (act == GEOGCN_ACT_TANH) { GEOGCN_ROWS(GEOGCN_ACT_TANH); }
        else if (act == GEOGCN_ACT_SIGMOID) { GEOGCN_ROWS(GEOGCN_ACT_SIGMOID); }
        else { GEOGCN_ROWS(GEOGCN_ACT_NONE); }
        GEOGCN_LAUNCH_CHECK("spmm_rows_kernel");
    }
#undef GEOGCN_ROWS
#undef GEOGCN_ROWS__
    if (n_chunks > 0) {
        const int Fpad = (int)std::min<int64_t>(ldc, ldp);
        const dim3 rgrid((unsigned)plan->n_long, (unsigned)cdiv(Fpad, kWave));
#define GEOGCN_RED(ACT) GEOGCN_RED_(ACT, 0)
#define GEOGCN_RED_(ACT, HW_)                                                                    \
    hipLaunchKernelGGL((spmm_long_reduce_kernel<ACT, HW_>), rgrid, dim3(kBlock), 0, st,          \
                       plan->d_long_rows, plan->d_long_first, ws, ldp, C, ldc, F, Fpad, bias, hw)
        if (hw.T) GEOGCN_RED_(GEOGCN_ACT_TANH, 1);
        else if (act == GEOGCN_ACT_TANH) GEOGCN_RED(GEOGCN_ACT_TANH);
        else if (act == GEOGCN_ACT_SIGMOID) GEOGCN_RED(GEOGCN_ACT_SIGMOID);
        else GEOGCN_RED(GEOGCN_ACT_NONE);
#undef GEOGCN_RED
#undef GEOGCN_RED_
        GEOGCN_LAUNCH_CHECK("spmm_long_reduce_kernel");
        // softmax epilogue: the long rows' logits (bias added by the combine) sit in C; their softmax in place, row by row
        if (hw.softmax)
            if (const int rc = softmax_rows_indexed_launch(plan->d_long_rows, plan->n_long, F, C, ldc, hw.amax, st)) return rc;
    }
    if (timed) {        // the pair brackets the WHOLE product: row kernel + the long rows' ordered combine
        GEOGCN_HIP(hipEventRecord(tm->end[tm->used], st));
        tm->used++;
    }
    return 0;
}

}  // namespace
}  // namespace geogcn

namespace geogcn {
namespace {

template <int BF>
int spmm_csr_impl(const char* fn, const geogcn_spmm_plan* plan, int32_t n_rows, int32_t n_cols, int64_t nnz,
                  const int32_t* rowptr, const int32_t* colidx, const float* val, const void* B, int64_t ldb,
                  float* C, int64_t ldc, int32_t F, const float* bias, int32_t act, void* ws, size_t ws_bytes,
                  void* stream, const HwArgs hw = HwArgs{nullptr, nullptr, nullptr, 0, nullptr, 0}) {
    GEOGCN_REQUIRE(n_rows >= 0 && n_cols >= 0 && nnz >= 0 && F >= 0, GEOGCN_E_SIZE,
                   "%s: negative size", fn);
    if (n_rows == 0 || F == 0) return 0;
    GEOGCN_REQUIRE(rowptr && C && (nnz == 0 || (colidx && val && B)), GEOGCN_E_NULL,
                   "%s: null pointer", fn);
    GEOGCN_REQUIRE(ldb >= F && ldc >= F, GEOGCN_E_SIZE, "%s: ld < F (ldb=%lld ldc=%lld F=%d)", fn,
                   (long long)ldb, (long long)ldc, F);
    GEOGCN_REQUIRE(act >= GEOGCN_ACT_NONE && act <= GEOGCN_ACT_SIGMOID, GEOGCN_E_ARG,
                   "%s: unknown act %d", fn, act);
    GEOGCN_REQUIRE(!plan || plan->n_rows == n_rows, GEOGCN_E_ARG,
                   "%s: plan built for %d rows, called with %d", fn, plan ? plan->n_rows : 0,
                   n_rows);
    hipStream_t st = (hipStream_t)stream;
    const int F4 = (F + 3) / 4;
    if constexpr (BF) {
        // bf16 rows are read in 16-byte pieces of 8 features: 16-byte row starts, pitch covering whole pieces
        const int F8 = (F + 7) / 8;
        GEOGCN_REQUIRE(ldb % 8 == 0 && ldb >= (int64_t)F8 * 8 && ldc % 4 == 0 && ldc >= (int64_t)F4 * 4 && aligned16(B) &&
                           aligned16(C) && F <= 1024,
                       GEOGCN_E_ALIGN,
                       "%s: needs 16-byte aligned B and C, ldb %% 8 == 0 and >= roundup8(F), ldc %% 4 == 0, F <= 1024 "
                       "(ldb=%lld ldc=%lld F=%d)", fn, (long long)ldb, (long long)ldc, F);
        const size_t need = geogcn_spmm_workspace_bytes(plan, F);
        GEOGCN_REQUIRE(need == 0 || (ws && ws_bytes >= need && aligned16(ws)), GEOGCN_E_ARG,
                       "%s: workspace too small or misaligned (%zu < %zu)", fn, ws_bytes, need);
        float* wsf = (float*)ws;
        // 8 lanes per row while that keeps <= 5 passes (F <= 320: every lane of a wave fetches 16 useful bytes,
        // 8 rows in flight per wave); 16 lanes per row beyond
#define GEOGCN_BF(K16, G_)                                                                                       \
    return launch_k4<2 * K16, G_, 1>(plan, n_rows, rowptr, colidx, val, B, ldb, C, ldc, F, bias, act, wsf, st, nnz, hw)
        const int k8 = (F8 + 7) / 8, k16 = (F8 + 15) / 16;
        if (k8 <= 5) {
            switch (k8) {
                case 1: GEOGCN_BF(1, 8);
                case 2: GEOGCN_BF(2, 8);
                case 3: GEOGCN_BF(3, 8);
                case 4: GEOGCN_BF(4, 8);
                default: GEOGCN_BF(5, 8);
            }
        }
        switch (k16) {
            case 3: GEOGCN_BF(3, 16);
            case 4: GEOGCN_BF(4, 16);
            case 5: GEOGCN_BF(5, 16);
            case 6: GEOGCN_BF(6, 16);
            case 7: GEOGCN_BF(7, 16);
            default: GEOGCN_BF(8, 16);
        }
#undef GEOGCN_BF
    } else {
    const bool vec_ok = (ldb % 4 == 0) && (ldc % 4 == 0) && aligned16(B) && aligned16(C) &&
                        ldb >= (int64_t)F4 * 4 && ldc >= (int64_t)F4 * 4 && F4 <= 16 * 16;      // F <= 1024
    if (!vec_ok) {
        GEOGCN_REQUIRE(!hw.T && !hw.init, GEOGCN_E_ALIGN, "%s: the highway / accumulate forms need float4-addressable operands", fn);
        const dim3 grid((unsigned)cdiv(n_rows, kBlock / kWave));
#define GEOGCN_SC(ACT)                                                                              \
    hipLaunchKernelGGL((spmm_scalar_kernel<ACT>), grid, dim3(kBlock), 0, st, n_rows, rowptr, colidx, \
                       val, (const float*)B, ldb, C, ldc, F, bias)
        if (act == GEOGCN_ACT_TANH) GEOGCN_SC(GEOGCN_ACT_TANH);
        else if (act == GEOGCN_ACT_SIGMOID) GEOGCN_SC(GEOGCN_ACT_SIGMOID);
        else GEOGCN_SC(GEOGCN_ACT_NONE);
#undef GEOGCN_SC
        GEOGCN_LAUNCH_CHECK("spmm_scalar_kernel");
        return 0;
    }
    const size_t need = geogcn_spmm_workspace_bytes(plan, F);
    GEOGCN_REQUIRE(need == 0 || (ws && ws_bytes >= need && aligned16(ws)), GEOGCN_E_ARG,
                   "%s: workspace too small or misaligned (%zu < %zu)", fn, ws_bytes, need);
    float* wsf = (float*)ws;
    // narrow operands (F <= 32, e.g. the per-rank feature panels of C = 256 over 8 GPUs): 8 lanes per row,
    // twice as many rows in flight per wave (measured 0.306 -> 0.267 ms at F = 16); otherwise 16 lanes per row
    if (F4 <= 8) return launch_k4<1, 8, 0>(plan, n_rows, rowptr, colidx, val, B, ldb, C, ldc, F, bias, act, wsf, st, nnz, hw);
    // (36..64 columns -- the per-rank feature panel of a 300-wide layer over 8 GPUs is 40 -- stay on 16 lanes x 1 float4:
    //  8 lanes x 2 float4, eight rows in flight per wave, measured slower: 0.411 vs 0.385 ms at F = 40.  What sets the
    //  time of these narrow products is the number of 128-byte lines per gathered row: F = 32 0.25 ms, F = 40 0.39, F = 64 0.41)
    const int K4 = (F4 + kGroup - 1) / kGroup;
    switch (K4) {
#define GEOGCN_CASE(K)                                                                             \
    case K:                                                                                        \
        return launch_k4<K, kGroup, 0>(plan, n_rows, rowptr, colidx, val, B, ldb, C, ldc, F, bias, act, wsf, st, nnz, hw);
        GEOGCN_CASE(1)
        GEOGCN_CASE(2)
        GEOGCN_CASE(3)
        GEOGCN_CASE(4)
        GEOGCN_CASE(5)
        GEOGCN_CASE(6)
        GEOGCN_CASE(7)
        GEOGCN_CASE(8)
        GEOGCN_CASE(9)
        GEOGCN_CASE(10)
        GEOGCN_CASE(11)     // hid 700..1024: the WORLD configuration of the reference uses 900 (README.md:180)
        GEOGCN_CASE(12)
        GEOGCN_CASE(13)
        GEOGCN_CASE(14)
        GEOGCN_CASE(15)
        GEOGCN_CASE(16)
#undef GEOGCN_CASE
        default:
            break;
    }
    set_error("%s: F=%d not supported", fn, F);
    return GEOGCN_E_ARG;
    }   // fp32 operand
}


}  // namespace
}  // namespace geogcn

using namespace geogcn;

extern "C" {

int geogcn_timer_create(int32_t capacity, geogcn_timer** out) {
    GEOGCN_REQUIRE(out && capacity > 0, GEOGCN_E_ARG, "timer_create: bad arguments");
    auto* t = new geogcn_timer();
    t->begin.resize(capacity);
    t->end.resize(capacity);
    for (int i = 0; i < capacity; ++i) {
        hipError_t e = hipEventCreate(&t->begin[i]);
        if (e == hipSuccess) e = hipEventCreate(&t->end[i]);
        if (e != hipSuccess) {
            set_error("timer_create: %s", hipGetErrorString(e));
            delete t;
            return (int)e;
        }
    }
    *out = t;
    return 0;
}

void geogcn_timer_destroy(geogcn_timer* t) {
    if (!t) return;
    for (auto& e : t->begin) (void)hipEventDestroy(e);
    for (auto& e : t->end) (void)hipEventDestroy(e);
    delete t;
}

int geogcn_spmm_plan_attach_timer(geogcn_spmm_plan* plan, geogcn_timer* t, int32_t only_F) {
    GEOGCN_REQUIRE(plan, GEOGCN_E_NULL, "spmm_plan_attach_timer: null plan");
    plan->timer = t;
    plan->timer_F = only_F;
    if (t) t->used = 0;
    return 0;
}

int geogcn_timer_read_ms(geogcn_timer* t, float* out_ms, int32_t max_out, int32_t* n_out) {
    GEOGCN_REQUIRE(t && out_ms && n_out, GEOGCN_E_NULL, "timer_read_ms: null pointer");
    const int n = std::min<int>(t->used, max_out);
    for (int i = 0; i < n; ++i) {
        GEOGCN_HIP(hipEventSynchronize(t->end[i]));
        GEOGCN_HIP(hipEventElapsedTime(&out_ms[i], t->begin[i], t->end[i]));
    }
    *n_out = n;
    return 0;
}

int geogcn_spmm_plan_create(int32_t n_rows, const int32_t* rowptr_host, int32_t long_row_nnz, int32_t chunk_nnz,
                            int32_t chunks_with_owner, geogcn_spmm_plan** out) {
    GEOGCN_REQUIRE(rowptr_host && out, GEOGCN_E_NULL, "spmm_plan_create: null pointer");
    GEOGCN_REQUIRE(n_rows >= 0 && long_row_nnz > 0 && chunk_nnz > 0, GEOGCN_E_SIZE,
                   "spmm_plan_create: bad sizes n_rows=%d long=%d chunk=%d", n_rows, long_row_nnz,
                   chunk_nnz);
    std::vector<int> long_rows, long_first, cs, ce;
    for (int r = 0; r < n_rows; ++r) {
        const int s = rowptr_host[r], e = rowptr_host[r + 1];
        GEOGCN_REQUIRE(e >= s, GEOGCN_E_SIZE, "spmm_plan_create: rowptr not monotone at row %d", r);
        if (e - s > long_row_nnz) {
            long_rows.push_back(r);
            long_first.push_back((int)cs.size());
            for (int p = s; p < e; p += chunk_nnz) {
                const int pe = std::min(e, p + chunk_nnz);
                cs.push_back(p);
                ce.push_back(pe);
            }
        }
    }
    long_first.push_back((int)cs.size());
    auto* plan = new geogcn_spmm_plan();
    {
        // row ranges of the 8 XCDs: cut where the running count of stored entries + a per-row term crosses k/8 of the
        // total, rounded to 64 rows (= whole row blocks for 8- and 16-lane groups).  Long rows count in full when their
        // chunks run on the XCD that owns the row, not at all when the chunks are dealt round the XCDs.
        auto cost = [&](int r) -> int64_t {
            const int64_t nz = rowptr_host[r + 1] - rowptr_host[r];
            return ((nz > long_row_nnz && !chunks_with_owner) ? 0 : nz) + 4;
        };
        int64_t total = 0;
        for (int r = 0; r < n_rows; ++r) total += cost(r);
        int64_t run = 0;
        int x = 1;
        for (int r = 0; r < n_rows && x < 8; ++r) {
            run += cost(r);
            while (x < 8 && run * 8 >= total * x) {
                plan->xcd_lo[x] = std::min(n_rows, (r + 1 + 63) / 64 * 64);
                ++x;
            }
        }
        for (; x < 8; ++x) plan->xcd_lo[x] = n_rows;
        plan->xcd_lo[8] = n_rows;
        for (int k = 1; k <= 8; ++k) plan->xcd_lo[k] = std::max(plan->xcd_lo[k], plan->xcd_lo[k - 1]);
    }
    // Per group width, the XCDs' block schedules.  Chunks were numbered in row order above; chunk block c = chunks
    // [c * gpb, (c + 1) * gpb).
    //   chunks_with_owner = 1 (a numbering with locality): a chunk block runs on the XCD that owns the row of its first chunk,
    //     slotted in just before the row block that row is in -- the hub's neighbours are then in the same L2 window as the
    //     rows around it (community graph + label-propagation order: 1.084 -> 1.018 ms);
    //   chunks_with_owner = 0 (no locality to keep): chunk block c goes to XCD c % 8 and leads that XCD's schedule, so the
    //     latency-bound 128-entry walks start first and are spread over the chip.  On the pinned power-law graph the hubs are
    //     the lowest-numbered rows: with the owner rule ONE XCD would walk nearly all chunks while the other seven share the
    //     short rows (measured: the highway-fused forward product 2.12 -> 2.34 ms).
    std::vector<int> sched[2];
    if (!cs.empty()) {
        std::vector<int> chunk_row(cs.size());
        for (size_t lr = 0; lr < long_rows.size(); ++lr)
            for (int c = long_first[lr]; c < long_first[lr + 1]; ++c) chunk_row[c] = long_rows[lr];
        for (int w = 0; w < 2; ++w) {
            const int gpb = kRowBlock / (w ? 8 : 16);
            const int n_cb = (int)cdiv((int64_t)cs.size(), gpb);
            std::vector<std::vector<int>> lists(kNumXCD);
            if (!chunks_with_owner)
                for (int c = 0; c < n_cb; ++c) lists[c % kNumXCD].push_back(-1 - c);
            size_t slots = 0;
            int c = 0;
            for (int x = 0; x < kNumXCD; ++x) {
                const int lo = plan->xcd_lo[x], hi = plan->xcd_lo[x + 1];
                const int rb = (int)cdiv(hi - lo, gpb);
                for (int j = 0; j < rb; ++j) {
                    if (chunks_with_owner)
                        while (c < n_cb && chunk_row[(size_t)c * gpb] < lo + (j + 1) * gpb) lists[x].push_back(-1 - c++);
                    lists[x].push_back(j);
                }
                if (chunks_with_owner && x == kNumXCD - 1)
                    while (c < n_cb) lists[x].push_back(-1 - c++);
                slots = std::max(slots, lists[x].size());
            }
            plan->sched_slots[w] = (int)slots;
            sched[w].assign(slots * kNumXCD, kIdleSlot);
            for (int x = 0; x < kNumXCD; ++x)
                for (size_t k = 0; k < lists[x].size(); ++k) sched[w][k * kNumXCD + x] = lists[x][k];
        }
    }
    plan->n_rows = n_rows;
    plan->long_row_nnz = long_row_nnz;
    plan->chunk_nnz = chunk_nnz;
    plan->n_long = (int64_t)long_rows.size();
    plan->n_chunks = (int64_t)cs.size();
    auto upload = [](const std::vector<int>& v, int** d) -> hipError_t {
        if (v.empty()) return hipSuccess;
        hipError_t e = hipMalloc((void**)d, v.size() * sizeof(int));
        if (e != hipSuccess) return e;
        return hipMemcpy(*d, v.data(), v.size() * sizeof(int), hipMemcpyHostToDevice);
    };
    if (plan->n_long > 0) {
        hipError_t e = upload(long_rows, &plan->d_long_rows);
        if (e == hipSuccess) e = upload(long_first, &plan->d_long_first);
        if (e == hipSuccess) e = upload(cs, &plan->d_chunk_start);
        if (e == hipSuccess) e = upload(ce, &plan->d_chunk_end);
        if (e == hipSuccess) e = upload(sched[0], &plan->d_sched[0]);
        if (e == hipSuccess) e = upload(sched[1], &plan->d_sched[1]);
        if (e != hipSuccess) {
            set_error("spmm_plan_create: %s", hipGetErrorString(e));
            geogcn_spmm_plan_destroy(plan);
            return (int)e;
        }
    }
    *out = plan;
    return 0;
}

void geogcn_spmm_plan_destroy(geogcn_spmm_plan* plan) {
    if (!plan) return;
    if (plan->d_long_rows) (void)hipFree(plan->d_long_rows);
    if (plan->d_long_first) (void)hipFree(plan->d_long_first);
    if (plan->d_chunk_start) (void)hipFree(plan->d_chunk_start);
    if (plan->d_chunk_end) (void)hipFree(plan->d_chunk_end);
    if (plan->d_sched[0]) (void)hipFree(plan->d_sched[0]);
    if (plan->d_sched[1]) (void)hipFree(plan->d_sched[1]);
    delete plan;
}

int64_t geogcn_spmm_plan_num_long_rows(const geogcn_spmm_plan* plan) { return plan ? plan->n_long : 0; }
int64_t geogcn_spmm_plan_num_chunks(const geogcn_spmm_plan* plan) { return plan ? plan->n_chunks : 0; }

size_t geogcn_spmm_workspace_bytes(const geogcn_spmm_plan* plan, int32_t F) {
    if (!plan || plan->n_chunks == 0 || F <= 0) return 0;
    return (size_t)plan->n_chunks * (size_t)((F + 3) / 4) * 4 * sizeof(float);
}

int geogcn_spmm_csr_f32(const geogcn_spmm_plan* plan, int32_t n_rows, int32_t n_cols, int64_t nnz,
                        const int32_t* rowptr, const int32_t* colidx, const float* val,
                        const float* B, int64_t ldb, float* C, int64_t ldc, int32_t F,
                        const float* bias, int32_t act, void* ws, size_t ws_bytes, void* stream) {
    return spmm_csr_impl<0>("spmm_csr_f32", plan, n_rows, n_cols, nnz, rowptr, colidx, val, B, ldb, C, ldc, F, bias,
                            act, ws, ws_bytes, stream);
}

// Hc = tanh(A.B + bias) and Hout = T*Hc + (1-T)*H in one launch; `b_bf16` selects the bf16 gathered operand
int geogcn_spmm_csr_highway_f32(const geogcn_spmm_plan* plan, int32_t n_rows, int32_t n_cols, int64_t nnz,
                                const int32_t* rowptr, const int32_t* colidx, const float* val, const void* B,
                                int64_t ldb, int32_t b_bf16, int32_t F, const float* bias, const float* T,
                                const float* H, int64_t ld, float* Hc, float* Hout, void* ws, size_t ws_bytes,
                                void* stream) {
    GEOGCN_REQUIRE(T && H && Hc && Hout, GEOGCN_E_NULL, "spmm_csr_highway_f32: null pointer");
    GEOGCN_REQUIRE(ld % 4 == 0 && ld >= (int64_t)((F + 3) / 4) * 4 && aligned16(T) && aligned16(H) && aligned16(Hout),
                   GEOGCN_E_ALIGN, "spmm_csr_highway_f32: T, H, Hc, Hout need 16-byte bases and a pitch %% 4 == 0 (ld=%lld)",
                   (long long)ld);
    const HwArgs hw{T, H, Hout, ld, nullptr, 0};
    if (b_bf16)
        return spmm_csr_impl<1>("spmm_csr_highway_f32", plan, n_rows, n_cols, nnz, rowptr, colidx, val, B, ldb, Hc, ld, F,
                                bias, GEOGCN_ACT_TANH, ws, ws_bytes, stream, hw);
    return spmm_csr_impl<0>("spmm_csr_highway_f32", plan, n_rows, n_cols, nnz, rowptr, colidx, val, B, ldb, Hc, ld, F,
                            bias, GEOGCN_ACT_TANH, ws, ws_bytes, stream, hw);
}

// P = softmax(A.B + bias) row by row [+ the first index of each row's maximum]: the output layer's product and its
// nonlinearity in one launch -- the logits are never written
int geogcn_spmm_csr_softmax_f32(const geogcn_spmm_plan* plan, int32_t n_rows, int32_t n_cols, int64_t nnz,
                                const int32_t* rowptr, const int32_t* colidx, const float* val, const float* B, int64_t ldb,
                                float* P, int64_t ldp, int32_t F, const float* bias, int32_t* argmax_out, void* ws,
                                size_t ws_bytes, void* stream) {
    const int F4 = (F + 3) / 4;
    GEOGCN_REQUIRE(F >= 0 && (F == 0 || (F4 > 8 && F4 <= 8 * kGroup)), GEOGCN_E_ARG,
                   "spmm_csr_softmax_f32: F=%d outside (32, %d] (narrower / wider outputs: geogcn_spmm_csr_f32 + geogcn_softmax_rows_f32)", F,
                   32 * kGroup);
    GEOGCN_REQUIRE(ldb % 4 == 0 && ldp % 4 == 0 && aligned16(B) && aligned16(P) && ldb >= (int64_t)F4 * 4 && ldp >= (int64_t)F4 * 4,
                   GEOGCN_E_ALIGN, "spmm_csr_softmax_f32: needs 16-byte aligned B and P and pitches %% 4 == 0, >= roundup4(F)");
    HwArgs hw{nullptr, nullptr, nullptr, 0, nullptr, 0};
    hw.amax = argmax_out;
    hw.softmax = 1;
    return spmm_csr_impl<0>("spmm_csr_softmax_f32", plan, n_rows, n_cols, nnz, rowptr, colidx, val, B, ldb, P, ldp, F, bias,
                            GEOGCN_ACT_NONE, ws, ws_bytes, stream, hw);
}

// C = act(C + A.B + bias): the accumulators start from the row already in C
int geogcn_spmm_csr_acc_f32(const geogcn_spmm_plan* plan, int32_t n_rows, int32_t n_cols, int64_t nnz,
                            const int32_t* rowptr, const int32_t* colidx, const float* val, const float* B, int64_t ldb,
                            float* C, int64_t ldc, int32_t F, const float* bias, int32_t act, void* ws, size_t ws_bytes,
                            void* stream) {
    const HwArgs hw{nullptr, nullptr, nullptr, 0, C, ldc};
    return spmm_csr_impl<0>("spmm_csr_acc_f32", plan, n_rows, n_cols, nnz, rowptr, colidx, val, B, ldb, C, ldc, F, bias,
                            act, ws, ws_bytes, stream, hw);
}

int geogcn_spmm_csr_bf16b(const geogcn_spmm_plan* plan, int32_t n_rows, int32_t n_cols, int64_t nnz,
                          const int32_t* rowptr, const int32_t* colidx, const float* val,
                          const uint16_t* B, int64_t ldb, float* C, int64_t ldc, int32_t F,
                          const float* bias, int32_t act, void* ws, size_t ws_bytes, void* stream) {
    return spmm_csr_impl<1>("spmm_csr_bf16b", plan, n_rows, n_cols, nnz, rowptr, colidx, val, B, ldb, C, ldc, F,
                            bias, act, ws, ws_bytes, stream);
}

}  // extern "C"
