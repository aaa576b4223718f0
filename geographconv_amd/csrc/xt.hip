// dW0 = X^T . dS0 for a bag-of-words X -- the gradient Theano derives for structured_dot(X, W0)
// (reference gcnmodel.py:39; StructuredDot grad w.r.t. the dense operand), gfx950.
//
// Why not the plain row gather over CSR(X^T): a vocabulary row references thousands of documents spread over all of
// dS0 (528 MB at the TwitterUS shape), so every gathered 1.2 KB row comes from beyond the 4 MB L2 and is fetched ~26
// times in total (14.6 GB of fabric traffic for 11.4 M tail nonzeros: the kernel runs at the ~7 TB/s beyond-L2 gather
// ceiling).  tools/micro/xcd_window.hip shows what the chip does when the gathers of one XCD stay inside a sliding
// window of <= 2.6 MB: 26-29 TB/s.  So the DOCUMENTS are partitioned and swept, not gathered at random:
//   * the grid is 8 "virtual XCDs" x 32 workgroups of 1024 threads, one per CU (block b runs on XCD b % 8 -- measured,
//     a speed assumption only: nothing below depends on it for correctness);  virtual XCD x owns the contiguous
//     document range [doc_lo[x], doc_lo[x+1]) and all its workgroups walk it in blocks of `doc_block` rows
//     (2048 rows = 2.6 MB of dS0), held together by a barrier per block inside a workgroup and by doing equal work;
//   * the unit of work is (vocabulary row w, part p of P): part p takes the stored entries of w whose position is
//     = p (mod P).  P grows with the row's nonzero count so that no unit is heavier than the average 16-lane group's
//     share, and the units are dealt to (batch, workgroup, group) slots by a snake over their weights: every group owns
//     at most 2 units per batch, equal counts and near-equal nonzeros -- a block takes every group about the same time;
//   * a group keeps its units' accumulators (K4 float4 per lane and unit) and cursors in REGISTERS across the sweep;
//     per block it loads the next <= 16 entries of each unit (both loads issued before either is used), finds by ballot
//     how many fall inside the block (documents are sorted inside a row) and gathers those rows of dS0 -- L2 hits --
//     with sequential fmaf in stored order;
//   * at the end of a batch the accumulators go to partial[x][unit]; a second kernel adds, for every vocabulary row,
//     its parts and their 8 per-XCD partials in fixed order and writes dW (rows without nonzeros: zeros).
// No atomics, fixed orders everywhere => bitwise reproducible.  HBM traffic: dS0 once per batch (3 batches at
// V = 10k, F = 300) + CSR(X^T) once per batch.
#include "common.h"

#include <stdlib.h>

#include <algorithm>
#include <numeric>
#include <vector>

namespace geogcn {
namespace {

constexpr int kGroup = 16;
constexpr int kBlock = 256;              // combine kernel
constexpr int kSlots = 32;               // workgroups per virtual XCD (one per CU)
constexpr int kUnitsPerGroup = 2;
// narrow rows (K4 <= 5: 2 x K4 float4 accumulators fit 128 VGPRs) run 1024-thread workgroups, wide ones 512 / 256
constexpr int wg_threads(int K4) { return K4 <= 5 ? 1024 : (K4 <= 10 ? 512 : 256); }
// documents per sweep block: 2048 rows of dS0 = 2.6 MB at F = 300, inside one XCD's 4 MB L2 (512 / 1024 / 3072-8192 rows
// measured slower, profiles/r02_xt_sweep.txt)
constexpr int kDocBlock = 2048;
// how many average group shares one unit may weigh: 1 = best balance but 4 batches at the TwitterUS shape, 2 = 3 batches
// (measured 1.50 against 1.59 ms)
constexpr double kUnitCap = 2.0;

typedef float f32x4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float4 ld4(const float* row, int q) {
    const f32x4v v = *(reinterpret_cast<const f32x4v*>(row) + q);
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void fma4(float4& acc, float a, const float4& b) {
    acc.x = fmaf(a, b.x, acc.x);
    acc.y = fmaf(a, b.y, acc.y);
    acc.z = fmaf(a, b.z, acc.z);
    acc.w = fmaf(a, b.w, acc.w);
}

struct XtArgs {
    const int* docidx;          // CSR(X^T): column = document
    const float* val;
    const float* G; int64_t ldg;
    int F;
    const int* unit_word;       // [n_units_padded] vocabulary row of the unit in slot i (-1: empty slot)
    const int* unit_part;       // [n_units_padded] p | (P << 16)
    const int* wptr;            // [9][n_words]: first nonzero of word w with doc >= doc_lo[x]
    const int* doc_lo;          // [9]
    int n_words, n_batches, doc_block;
    float* partial; int64_t ldp;        // [8][n_slots][ldp],  n_slots = n_batches * kSlots * groups * kUnitsPerGroup
    int64_t n_slots;
};

template <int K4>
__global__ __launch_bounds__(wg_threads(K4), 1) void xt_tail_kernel(const XtArgs a) {
    constexpr int kGroupsPerBlock = wg_threads(K4) / kGroup;
    constexpr int U = kUnitsPerGroup;
    const int x = blockIdx.x % kNumXCD, c = blockIdx.x / kNumXCD;
    const int g = threadIdx.x / kGroup, lane = threadIdx.x % kGroup;
    const int gshift = (threadIdx.x & 63) / kGroup * kGroup;        // my group's bits inside the wave ballot
    const int nF4 = (a.F + 3) >> 2;
    const int d0 = a.doc_lo[x], d1 = a.doc_lo[x + 1];
    for (int r = 0; r < a.n_batches; ++r) {
        // slot of (batch r, workgroup c, group g, unit u) -- the same for every virtual XCD
        const int64_t slot0 = (((int64_t)r * kSlots + c) * kGroupsPerBlock + g) * U;
        int cur[U], endp[U], part[U], npart[U];
        float4 acc[U][K4];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int w = a.unit_word[slot0 + u];
            const int pp = a.unit_part[slot0 + u];
            part[u] = pp & 0xffff;
            npart[u] = pp >> 16;
            cur[u] = endp[u] = 0;
            if (w >= 0) {
                cur[u] = a.wptr[(int64_t)x * a.n_words + w];
                endp[u] = a.wptr[(int64_t)(x + 1) * a.n_words + w];
            }
#pragma unroll
            for (int k = 0; k < K4; ++k) acc[u][k] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        // the next <= 16 entries of every unit are always loaded one step ahead (at the end of the previous block)
        int doc[U];
        float v[U];
        auto load_entries = [&](int u) {
            const int p = cur[u] + lane;
            doc[u] = 0x7fffffff;
            v[u] = 0.f;
            if (p < endp[u]) {
                doc[u] = a.docidx[p];
                v[u] = a.val[p];
            }
        };
#pragma unroll
        for (int u = 0; u < U; ++u) load_entries(u);
        int blk = 0;
        for (int b0 = d0; b0 < d1; b0 += a.doc_block, ++blk) {
            const int be = min(d1, b0 + a.doc_block);
            // the groups of a workgroup move through the document blocks together.  (Measured and removed in round 3: a
            // soft per-XCD rendezvous across workgroups -- no gain once the loads are balanced -- and an L2 prefetch of the
            // next block by dword touches -- 1.84-1.96 ms against 1.50: the touching wave itself waits ~2 us for HBM.)
            __syncthreads();
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (cur[u] >= endp[u]) continue;
                while (true) {
                    // documents are sorted inside a row: the entries of this block are a prefix of the 16 loaded
                    const unsigned long long m = __ballot(doc[u] < be);
                    const int cnt = __popc((unsigned)((m >> gshift) & 0xffffu));
                    // my part: positions = part (mod npart)
                    const int P = npart[u];
                    int t = part[u] - cur[u] % P;
                    if (t < 0) t += P;
                    for (; t + P < cnt; t += 2 * P) {
                        const int c0 = __shfl(doc[u], t, kGroup), c1 = __shfl(doc[u], t + P, kGroup);
                        const float a0 = __shfl(v[u], t, kGroup), a1 = __shfl(v[u], t + P, kGroup);
                        const float* r0 = a.G + (int64_t)c0 * a.ldg;
                        const float* r1 = a.G + (int64_t)c1 * a.ldg;
                        float4 v0[K4], v1[K4];
#pragma unroll
                        for (int k = 0; k < K4; ++k) {
                            const int q = lane + kGroup * k;
                            if (q < nF4) {
                                v0[k] = ld4(r0, q);
                                v1[k] = ld4(r1, q);
                            }
                        }
#pragma unroll
                        for (int k = 0; k < K4; ++k) {
                            const int q = lane + kGroup * k;
                            if (q < nF4) {
                                fma4(acc[u][k], a0, v0[k]);
                                fma4(acc[u][k], a1, v1[k]);
                            }
                        }
                    }
                    if (t < cnt) {
                        const int c0 = __shfl(doc[u], t, kGroup);
                        const float a0 = __shfl(v[u], t, kGroup);
                        const float* r0 = a.G + (int64_t)c0 * a.ldg;
#pragma unroll
                        for (int k = 0; k < K4; ++k) {
                            const int q = lane + kGroup * k;
                            if (q < nF4) fma4(acc[u][k], a0, ld4(r0, q));
                        }
                    }
                    cur[u] += cnt;
                    load_entries(u);              // the next 16 (for this block if the batch was full, else for the next)
                    if (cnt < kGroup) break;
                }
            }
        }
        // this batch's accumulators -> partial[x][slot]
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (a.unit_word[slot0 + u] < 0) continue;
            float4* out = reinterpret_cast<float4*>(a.partial + ((int64_t)x * a.n_slots + slot0 + u) * a.ldp);
#pragma unroll
            for (int k = 0; k < K4; ++k) {
                const int q = lane + kGroup * k;
                if (q < nF4) out[q] = acc[u][k];
            }
        }
    }
}

// dW[w][:] = sum over the parts of w (in part order), for each part over x (in XCD order), of partial[x][slot];
// rows without nonzeros (no slots) = 0
__global__ __launch_bounds__(kBlock) void xt_combine_kernel(int n_words, int F, int F4, const int* __restrict__ word_slot_start,
                                                            const int* __restrict__ word_slots,
                                                            const float* __restrict__ partial, int64_t ldp, int64_t n_slots,
                                                            float* __restrict__ dW, int64_t ldw) {
    const int64_t total = (int64_t)n_words * F4;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += (int64_t)gridDim.x * kBlock) {
        const int w = (int)(e / F4), q = (int)(e - (int64_t)w * F4);
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int i = word_slot_start[w]; i < word_slot_start[w + 1]; ++i) {
            const int64_t slot = word_slots[i];
#pragma unroll
            for (int x = 0; x < kNumXCD; ++x) {
                const float4 v = *reinterpret_cast<const float4*>(partial + ((int64_t)x * n_slots + slot) * ldp + q * 4);
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
        }
        float o[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (q * 4 + i >= F) o[i] = 0.f;
        *reinterpret_cast<float4*>(dW + (int64_t)w * ldw + q * 4) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

}  // namespace
}  // namespace geogcn

struct geogcn_xt_plan {
    int32_t n_words = 0, n_docs = 0, F = 0, K4 = 0;
    int32_t n_batches = 0, doc_block = 0, max_blocks = 0, groups = 0;
    int64_t n_slots = 0, n_units = 0, nnz = 0;
    int* d_unit_word = nullptr;
    int* d_unit_part = nullptr;
    int* d_wptr = nullptr;
    int* d_doc_lo = nullptr;
    int* d_word_slot_start = nullptr;
    int* d_word_slots = nullptr;
};

using namespace geogcn;

extern "C" {

int geogcn_xt_plan_create(int32_t n_words, int32_t n_docs, const int32_t* rowptr_t_host, const int32_t* docidx_t_host,
                          int32_t F, geogcn_xt_plan** out) {
    GEOGCN_REQUIRE(rowptr_t_host && out && (rowptr_t_host[n_words] == 0 || docidx_t_host), GEOGCN_E_NULL,
                   "xt_plan_create: null pointer");
    GEOGCN_REQUIRE(n_words >= 0 && n_docs >= 0 && F > 0 && F <= 1024, GEOGCN_E_SIZE, "xt_plan_create: bad sizes (F=%d)", F);
    auto* plan = new geogcn_xt_plan();
    plan->n_words = n_words;
    plan->n_docs = n_docs;
    plan->F = F;
    plan->K4 = (int)cdiv(cdiv(F, 4), kGroup);
    plan->nnz = rowptr_t_host[n_words];
    plan->doc_block = kDocBlock;
    plan->groups = wg_threads(plan->K4) / kGroup;
    const int64_t slots_per_batch = (int64_t)kSlots * plan->groups * kUnitsPerGroup;      // per virtual XCD
    std::vector<int> words;
    for (int w = 0; w < n_words; ++w) {
        GEOGCN_REQUIRE(rowptr_t_host[w + 1] >= rowptr_t_host[w], GEOGCN_E_SIZE, "xt_plan_create: rowptr not monotone at %d", w);
        // precondition of the sweep (the entries of a document block are a PREFIX of the next 16, and the per-XCD entry
        // points below are found by binary search): document indices ascending inside every row
        for (int p = rowptr_t_host[w] + 1; p < rowptr_t_host[w + 1]; ++p)
            GEOGCN_REQUIRE(docidx_t_host[p] > docidx_t_host[p - 1], GEOGCN_E_ARG,
                           "xt_plan_create: row %d of CSR(X^T) is not in strictly ascending document order", w);
        if (rowptr_t_host[w + 1] > rowptr_t_host[w]) words.push_back(w);
    }
    // units (word, part p of P): P so that no unit outweighs the average group's share of a batch.  The batch count
    // depends on the unit count and vice versa: iterate (converges in a few rounds).
    struct Unit { int w, p, P; int64_t load; };
    std::vector<Unit> units;
    int n_batches = (int)std::max<int64_t>(1, cdiv((int64_t)words.size(), slots_per_batch));
    for (int it = 0; it < 8; ++it) {
        const double share = kUnitCap * std::max(16.0, (double)plan->nnz / ((double)n_batches * kSlots * plan->groups));
        units.clear();
        for (int w : words) {
            const int64_t nz = rowptr_t_host[w + 1] - rowptr_t_host[w];
            const int P = (int)std::min<int64_t>(4096, std::max<int64_t>(1, (int64_t)((double)nz / share + 0.999)));
            for (int p = 0; p < P; ++p) units.push_back(Unit{w, p, P, nz / P});
        }
        const int need = (int)std::max<int64_t>(1, cdiv((int64_t)units.size(), slots_per_batch));
        if (need <= n_batches) break;
        n_batches = need;
    }
    plan->n_batches = n_batches;
    plan->n_units = (int64_t)units.size();
    plan->n_slots = (int64_t)n_batches * slots_per_batch;
    std::stable_sort(units.begin(), units.end(), [](const Unit& a, const Unit& b) { return a.load > b.load; });
    // deal: first one unit per (batch, workgroup, group) -- heaviest first, snake over all of them -- then the second
    // row in the opposite direction, so that the heaviest first unit is paired with the lightest second one
    const int64_t n_groups_total = (int64_t)n_batches * kSlots * plan->groups;
    std::vector<int> unit_word((size_t)plan->n_slots, -1), unit_part((size_t)plan->n_slots, 1 << 16);
    for (int64_t i = 0; i < (int64_t)units.size(); ++i) {
        const int64_t row = i / n_groups_total, k = i % n_groups_total;
        GEOGCN_REQUIRE(row < kUnitsPerGroup, GEOGCN_E_SIZE, "xt_plan_create: unit overflow");
        const int64_t gidx = (row & 1) ? n_groups_total - 1 - k : k;
        // interleave so that consecutive (similar-weight) units land in different batches / workgroups first
        const int64_t b = gidx % n_batches, rest = gidx / n_batches;
        const int64_t wg = rest % kSlots, grp = rest / kSlots;
        const int64_t slot = (((b * kSlots) + wg) * plan->groups + grp) * kUnitsPerGroup + row;
        unit_word[(size_t)slot] = units[(size_t)i].w;
        unit_part[(size_t)slot] = units[(size_t)i].p | (units[(size_t)i].P << 16);
    }
    // per word: its slots in part order
    std::vector<int> word_slot_start(n_words + 1, 0), word_slots(units.size());
    for (const Unit& u : units) word_slot_start[u.w + 1]++;
    for (int w = 0; w < n_words; ++w) word_slot_start[w + 1] += word_slot_start[w];
    for (int64_t sl = 0; sl < plan->n_slots; ++sl) {
        const int w = unit_word[(size_t)sl];
        if (w >= 0) word_slots[(size_t)(word_slot_start[w] + (unit_part[(size_t)sl] & 0xffff))] = (int)sl;
    }
    // document ranges of the 8 virtual XCDs (multiples of the block size except the last) + per-word entry points
    std::vector<int> doc_lo(kNumXCD + 1);
    const int64_t blocks = cdiv(n_docs, plan->doc_block);
    for (int x = 0; x <= kNumXCD; ++x) doc_lo[x] = (int)std::min<int64_t>(n_docs, (blocks * x / kNumXCD) * plan->doc_block);
    doc_lo[kNumXCD] = n_docs;
    plan->max_blocks = 1;
    for (int x = 0; x < kNumXCD; ++x)
        plan->max_blocks = std::max<int>(plan->max_blocks, (int)cdiv(doc_lo[x + 1] - doc_lo[x], plan->doc_block));
    std::vector<int> wptr((size_t)(kNumXCD + 1) * std::max(1, n_words), 0);
    for (int w = 0; w < n_words; ++w) {
        const int32_t* b = docidx_t_host + rowptr_t_host[w];
        const int32_t* e = docidx_t_host + rowptr_t_host[w + 1];
        for (int x = 0; x <= kNumXCD; ++x)
            wptr[(size_t)x * n_words + w] = rowptr_t_host[w] + (int)(std::lower_bound(b, e, doc_lo[x]) - b);
    }
    auto upload = [](const std::vector<int>& v, int** d) -> hipError_t {
        if (v.empty()) return hipSuccess;
        hipError_t e = hipMalloc((void**)d, v.size() * sizeof(int));
        if (e != hipSuccess) return e;
        return hipMemcpy(*d, v.data(), v.size() * sizeof(int), hipMemcpyHostToDevice);
    };
    hipError_t e = upload(unit_word, &plan->d_unit_word);
    if (e == hipSuccess) e = upload(unit_part, &plan->d_unit_part);
    if (e == hipSuccess) e = upload(wptr, &plan->d_wptr);
    if (e == hipSuccess) e = upload(doc_lo, &plan->d_doc_lo);
    if (e == hipSuccess) e = upload(word_slot_start, &plan->d_word_slot_start);
    if (e == hipSuccess) e = upload(word_slots, &plan->d_word_slots);
    if (e != hipSuccess) {
        set_error("xt_plan_create: %s", hipGetErrorString(e));
        geogcn_xt_plan_destroy(plan);
        return (int)e;
    }
    *out = plan;
    return 0;
}

void geogcn_xt_plan_destroy(geogcn_xt_plan* plan) {
    if (!plan) return;
    for (int* p : {plan->d_unit_word, plan->d_unit_part, plan->d_wptr, plan->d_doc_lo, plan->d_word_slot_start, plan->d_word_slots})
        if (p) (void)hipFree(p);
    delete plan;
}

size_t geogcn_xt_workspace_bytes(const geogcn_xt_plan* plan) {
    if (!plan || plan->n_units == 0) return 0;
    return (size_t)kNumXCD * (size_t)plan->n_slots * (size_t)(plan->K4 * kGroup * 4) * sizeof(float);
}

int geogcn_xt_dot_f32(const geogcn_xt_plan* plan, const int32_t* docidx_t, const float* val_t, const float* G, int64_t ldg,
                      float* dW, int64_t ldw, void* ws, size_t ws_bytes, void* stream) {
    GEOGCN_REQUIRE(plan && dW, GEOGCN_E_NULL, "xt_dot_f32: null pointer");
    if (plan->n_words == 0) return 0;
    const int F = plan->F, F4 = (F + 3) / 4;
    GEOGCN_REQUIRE(ldw % 4 == 0 && ldw >= (int64_t)F4 * 4 && aligned16(dW), GEOGCN_E_ALIGN, "xt_dot_f32: bad dW pitch / base");
    hipStream_t st = (hipStream_t)stream;
    float* partial = nullptr;
    const int64_t ldp = (int64_t)plan->K4 * kGroup * 4;
    if (plan->n_units > 0) {
        GEOGCN_REQUIRE(docidx_t && val_t && G, GEOGCN_E_NULL, "xt_dot_f32: null pointer");
        GEOGCN_REQUIRE(ldg % 4 == 0 && ldg >= (int64_t)F4 * 4 && aligned16(G), GEOGCN_E_ALIGN, "xt_dot_f32: bad G pitch / base");
        const size_t need = geogcn_xt_workspace_bytes(plan);
        GEOGCN_REQUIRE(ws && ws_bytes >= need && aligned16(ws), GEOGCN_E_ARG, "xt_dot_f32: workspace too small (%zu < %zu)",
                       ws_bytes, need);
        partial = (float*)ws;
        XtArgs a{docidx_t, val_t, G, ldg, F, plan->d_unit_word, plan->d_unit_part, plan->d_wptr, plan->d_doc_lo,
                 plan->n_words, plan->n_batches, plan->doc_block, partial, ldp, plan->n_slots};
        const dim3 grid((unsigned)(kNumXCD * kSlots));
        switch (plan->K4) {
#define GEOGCN_XT(K)                                                                \
    case K:                                                                         \
        hipLaunchKernelGGL(xt_tail_kernel<K>, grid, dim3(wg_threads(K)), 0, st, a); \
        break;
            GEOGCN_XT(1) GEOGCN_XT(2) GEOGCN_XT(3) GEOGCN_XT(4) GEOGCN_XT(5) GEOGCN_XT(6) GEOGCN_XT(7) GEOGCN_XT(8)
            GEOGCN_XT(9) GEOGCN_XT(10) GEOGCN_XT(11) GEOGCN_XT(12) GEOGCN_XT(13) GEOGCN_XT(14) GEOGCN_XT(15) GEOGCN_XT(16)
#undef GEOGCN_XT
            default:
                set_error("xt_dot_f32: F=%d not supported", F);
                return GEOGCN_E_ARG;
        }
        GEOGCN_LAUNCH_CHECK("xt_tail_kernel");
    }
    const int64_t total = (int64_t)plan->n_words * F4;
    const unsigned cgrid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(cdiv(total, kBlock), (int64_t)kNumCU * 8));
    hipLaunchKernelGGL(xt_combine_kernel, dim3(cgrid), dim3(kBlock), 0, st, plan->n_words, F, F4, plan->d_word_slot_start,
                       plan->d_word_slots, partial, ldp, plan->n_slots, dW, ldw);
    GEOGCN_LAUNCH_CHECK("xt_combine_kernel");
    return 0;
}

}  // extern "C"
