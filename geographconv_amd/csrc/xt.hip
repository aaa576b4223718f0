// dW0 = X^T . dS0 for a bag-of-words X -- the gradient Theano derives for structured_dot(X, W0)
// (reference gcnmodel.py:39; StructuredDot grad w.r.t. the dense operand), gfx950.
//
// Why not the plain row gather over CSR(X^T): each vocabulary row references ~N/V... thousands of documents
// spread over all of dS0 (528 MB at the TwitterUS shape), so every gathered 1.2 KB row comes from beyond the
// 4 MB L2 and is fetched ~18 times in total (9.8 GB of fabric traffic for 8.1 M nonzeros).  Here the DOCUMENTS
// are partitioned instead of streamed at random:
//   * the grid is 8 "virtual XCDs" x `slots` workgroups (block b runs on XCD b % 8 -- a speed assumption only,
//     nothing below depends on it for correctness);  virtual XCD x owns the contiguous document range
//     [doc_lo[x], doc_lo[x+1]) and sweeps it in blocks of `doc_block` rows (1024 rows = 1.3 MB of dS0: a third of
//     the XCD's L2, so that sweepers one or two blocks apart still share it);
//   * the vocabulary rows are dealt out to `slots * n_batches` bins, balanced by nonzeros AND by count;  in
//     batch r, slot c accumulates the rows of bin (r, c): every 16-lane group owns a fixed subset of the bin's
//     rows (row j -> group j % 16) and keeps one LDS accumulator row per owned word.  For each document
//     block, each owned word's nonzeros inside the block are gathered (L2 hits: all workgroups of the XCD are
//     inside the same ~2.5 MB window), summed in registers in stored (= document) order, and added to the
//     word's LDS row -- no atomics, no cross-group traffic, a fixed order => bitwise reproducible;
//   * at the end of a batch the LDS rows go to partial[x][word]; a second kernel adds the 8 partials of each
//     word in XCD order and writes dW (rows without nonzeros: zeros).
// HBM traffic: dS0 once per batch (n_batches = 3 at V = 10k, F = 300) + CSR(X^T) once.
#include "common.h"

#include <stdlib.h>

#include <algorithm>
#include <numeric>
#include <vector>

namespace geogcn {
namespace {

constexpr int kGroup = 16;
constexpr int kBlock = 256;              // combine kernel
// The sweep kernel runs ONE workgroup per CU (32 per virtual XCD): the fewer independent sweepers an XCD has, the
// closer together they stay inside the L2 window.  Narrow rows (K4 <= 6: <= 128 VGPRs) use 1024 threads = 64 groups,
// wide ones 512 threads.  Inside a workgroup the groups are held together by a barrier per document block.
constexpr int kSlots = 32;
constexpr int wg_threads(int K4) { return K4 <= 6 ? 1024 : 512; }
constexpr int kLdsBudget = 150 * 1024;
inline int doc_block_rows() {
    static const int v = [] {
        const char* e = getenv("GEOGCN_XT_DOC_BLOCK");
        const int b = e ? atoi(e) : 0;
        return b > 0 ? b : 1024;
    }();
    return v;
}

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
    const int* bin_start;       // [n_bins + 1] into bin_words
    const int* bin_words;       // vocabulary row of position p
    const int* wptr;            // [9][n_words]: first nonzero of word w with doc >= doc_lo[x]
    const int* doc_lo;          // [9]
    int n_words, slots, n_batches, cap, doc_block;
    float* partial; int64_t ldp;        // [8][n_pos][ldp]
    int64_t n_pos;
    // soft per-XCD rendezvous (nullable): arrive[(x * n_batches + r) * max_blocks + blk] counts the workgroups of
    // virtual XCD x that have ENTERED block blk of batch r.  A workgroup enters block blk + 2 only once all of them
    // have entered block blk -- with a bounded wait: results never depend on it, it only keeps the sweepers inside
    // a two-block window of dS0 so that the window stays in the XCD's L2.
    unsigned* arrive;
    int max_blocks, spin_limit;
};

template <int K4>
__global__ __launch_bounds__(wg_threads(K4), 1) void xt_tail_kernel(const XtArgs a) {
    constexpr int kGroupsPerBlock = wg_threads(K4) / kGroup;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float4* acc_lds = reinterpret_cast<float4*>(smem_raw);                       // [cap][K4 * 16]
    int* cur = reinterpret_cast<int*>(smem_raw + (size_t)a.cap * K4 * kGroup * sizeof(float4));      // [cap]
    int* endp = cur + a.cap;                                                     // [cap]
    const int x = blockIdx.x % kNumXCD, c = blockIdx.x / kNumXCD;
    const int g = threadIdx.x / kGroup, lane = threadIdx.x % kGroup;
    const int gshift = (threadIdx.x & 63) / kGroup * kGroup;        // my group's bits inside the wave ballot
    const int nF4 = (a.F + 3) >> 2;
    const int d0 = a.doc_lo[x], d1 = a.doc_lo[x + 1];
    for (int r = 0; r < a.n_batches; ++r) {
        const int bin = r * a.slots + c;
        const int ws = a.bin_start[bin], nw = a.bin_start[bin + 1] - ws;
        // my words: j = g, g + 16, ...  (only this group ever touches their LDS rows and cursors)
        for (int j = g; j < nw; j += kGroupsPerBlock) {
            const int w = a.bin_words[ws + j];
            if (lane == 0) {
                cur[j] = a.wptr[(int64_t)x * a.n_words + w];
                endp[j] = a.wptr[(int64_t)(x + 1) * a.n_words + w];
            }
#pragma unroll
            for (int k = 0; k < K4; ++k) acc_lds[(j * K4 + k) * kGroup + lane] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __syncthreads();
        int blk = 0;
        for (int b0 = d0; b0 < d1; b0 += a.doc_block, ++blk) {
            const int be = min(d1, b0 + a.doc_block);
            if (a.arrive) {
                if (threadIdx.x == 0) {
                    unsigned* base = a.arrive + ((int64_t)x * a.n_batches + r) * a.max_blocks;
                    __hip_atomic_fetch_add(base + blk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (blk >= 2) {
                        int spins = 0;
                        while (__hip_atomic_load(base + blk - 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)a.slots &&
                               ++spins < a.spin_limit)
                            __builtin_amdgcn_s_sleep(8);
                    }
                }
            }
            __syncthreads();          // the groups of a workgroup move through the document blocks together
            for (int j = g; j < nw; j += kGroupsPerBlock) {
                int s = cur[j];
                const int e = endp[j];
                if (s >= e) continue;
                const int s0 = s;
                float4 acc[K4];
#pragma unroll
                for (int k = 0; k < K4; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                while (true) {
                    const int p = s + lane;
                    int doc = 0x7fffffff;
                    float v = 0.f;
                    if (p < e) {
                        doc = a.docidx[p];
                        v = a.val[p];
                    }
                    // documents are sorted inside a row: the entries of this block are a prefix of the 16 loaded
                    const unsigned long long m = __ballot(doc < be);
                    const int cnt = __popc((unsigned)((m >> gshift) & 0xffffu));
                    int t = 0;
                    for (; t + 1 < cnt; t += 2) {
                        const int c0 = __shfl(doc, t, kGroup), c1 = __shfl(doc, t + 1, kGroup);
                        const float a0 = __shfl(v, t, kGroup), a1 = __shfl(v, t + 1, kGroup);
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
                                fma4(acc[k], a0, v0[k]);
                                fma4(acc[k], a1, v1[k]);
                            }
                        }
                    }
                    if (t < cnt) {
                        const int c0 = __shfl(doc, t, kGroup);
                        const float a0 = __shfl(v, t, kGroup);
                        const float* r0 = a.G + (int64_t)c0 * a.ldg;
#pragma unroll
                        for (int k = 0; k < K4; ++k) {
                            const int q = lane + kGroup * k;
                            if (q < nF4) fma4(acc[k], a0, ld4(r0, q));
                        }
                    }
                    s += cnt;
                    if (cnt < kGroup) break;
                }
                if (s != s0) {
#pragma unroll
                    for (int k = 0; k < K4; ++k) {
                        float4 o = acc_lds[(j * K4 + k) * kGroup + lane];
                        o.x += acc[k].x; o.y += acc[k].y; o.z += acc[k].z; o.w += acc[k].w;
                        acc_lds[(j * K4 + k) * kGroup + lane] = o;
                    }
                    if (lane == 0) cur[j] = s;
                }
            }
        }
        // this batch's rows -> partial[x][position]
        for (int j = g; j < nw; j += kGroupsPerBlock) {
            float4* out = reinterpret_cast<float4*>(a.partial + ((int64_t)x * a.n_pos + ws + j) * a.ldp);
#pragma unroll
            for (int k = 0; k < K4; ++k) {
                const int q = lane + kGroup * k;
                if (q < nF4) out[q] = acc_lds[(j * K4 + k) * kGroup + lane];
            }
        }
    }
}

// dW[w][:] = sum over x (in order) of partial[x][pos_of_word[w]][:]; rows without nonzeros (pos < 0) = 0
__global__ __launch_bounds__(kBlock) void xt_combine_kernel(int n_words, int F, int F4, const int* __restrict__ pos_of_word,
                                                            const float* __restrict__ partial, int64_t ldp, int64_t n_pos,
                                                            float* __restrict__ dW, int64_t ldw) {
    const int64_t total = (int64_t)n_words * F4;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += (int64_t)gridDim.x * kBlock) {
        const int w = (int)(e / F4), q = (int)(e - (int64_t)w * F4);
        const int p = pos_of_word[w];
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p >= 0) {
#pragma unroll
            for (int x = 0; x < kNumXCD; ++x) {
                const float4 v = *reinterpret_cast<const float4*>(partial + ((int64_t)x * n_pos + p) * ldp + q * 4);
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
    int32_t slots = 0, n_batches = 0, cap = 0, doc_block = 0, max_blocks = 0;
    int64_t n_pos = 0, nnz = 0;
    int* d_bin_start = nullptr;
    int* d_bin_words = nullptr;
    int* d_wptr = nullptr;
    int* d_doc_lo = nullptr;
    int* d_pos_of_word = nullptr;
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
    const int row_bytes = plan->K4 * kGroup * (int)sizeof(float4);
    plan->cap = std::max(1, (kLdsBudget - 1024) / (row_bytes + 8));
    plan->doc_block = doc_block_rows();
    // words with nonzeros, heaviest first
    std::vector<int> words;
    for (int w = 0; w < n_words; ++w) {
        GEOGCN_REQUIRE(rowptr_t_host[w + 1] >= rowptr_t_host[w], GEOGCN_E_SIZE, "xt_plan_create: rowptr not monotone at %d", w);
        if (rowptr_t_host[w + 1] > rowptr_t_host[w]) words.push_back(w);
    }
    std::stable_sort(words.begin(), words.end(), [&](int p, int q) {
        return rowptr_t_host[p + 1] - rowptr_t_host[p] > rowptr_t_host[q + 1] - rowptr_t_host[q];
    });
    const int64_t nw = (int64_t)words.size();
    plan->slots = kSlots;                                // x 8 virtual XCDs = 256 workgroups = one per CU
    plan->n_batches = (int)std::max<int64_t>(1, cdiv(nw, (int64_t)plan->slots * plan->cap));
    const int n_bins = plan->slots * plan->n_batches;
    // snake deal over the bins: equal counts (+-1) and near-equal nonzeros per bin
    std::vector<std::vector<int>> bins(n_bins);
    for (int64_t i = 0; i < nw; ++i) {
        const int64_t round = i / n_bins, k = i % n_bins;
        bins[(round & 1) ? n_bins - 1 - k : k].push_back(words[i]);
    }
    std::vector<int> bin_start(n_bins + 1, 0), bin_words;
    std::vector<int> pos_of_word(n_words, -1);
    for (int b = 0; b < n_bins; ++b) {
        GEOGCN_REQUIRE((int)bins[b].size() <= plan->cap, GEOGCN_E_SIZE, "xt_plan_create: bin overflow");
        bin_start[b] = (int)bin_words.size();
        for (int w : bins[b]) {
            pos_of_word[w] = (int)bin_words.size();
            bin_words.push_back(w);
        }
    }
    bin_start[n_bins] = (int)bin_words.size();
    plan->n_pos = (int64_t)bin_words.size();
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
    hipError_t e = upload(bin_start, &plan->d_bin_start);
    if (e == hipSuccess) e = upload(bin_words, &plan->d_bin_words);
    if (e == hipSuccess) e = upload(wptr, &plan->d_wptr);
    if (e == hipSuccess) e = upload(doc_lo, &plan->d_doc_lo);
    if (e == hipSuccess) e = upload(pos_of_word, &plan->d_pos_of_word);
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
    for (int* p : {plan->d_bin_start, plan->d_bin_words, plan->d_wptr, plan->d_doc_lo, plan->d_pos_of_word})
        if (p) (void)hipFree(p);
    delete plan;
}

static size_t xt_arrive_bytes(const geogcn_xt_plan* plan) {
    return (((size_t)kNumXCD * plan->n_batches * plan->max_blocks * sizeof(unsigned)) + 255) & ~(size_t)255;
}

size_t geogcn_xt_workspace_bytes(const geogcn_xt_plan* plan) {
    if (!plan || plan->n_pos == 0) return 0;
    return xt_arrive_bytes(plan) + (size_t)kNumXCD * (size_t)plan->n_pos * (size_t)(plan->K4 * kGroup * 4) * sizeof(float);
}

int geogcn_xt_dot_f32(const geogcn_xt_plan* plan, const int32_t* docidx_t, const float* val_t, const float* G, int64_t ldg,
                      float* dW, int64_t ldw, void* ws, size_t ws_bytes, void* stream) {
    GEOGCN_REQUIRE(plan && dW, GEOGCN_E_NULL, "xt_dot_f32: null pointer");
    if (plan->n_words == 0) return 0;
    const int F = plan->F, F4 = (F + 3) / 4;
    GEOGCN_REQUIRE(ldw % 4 == 0 && ldw >= (int64_t)F4 * 4 && aligned16(dW), GEOGCN_E_ALIGN, "xt_dot_f32: bad dW pitch / base");
    hipStream_t st = (hipStream_t)stream;
    float* partial = nullptr;
    if (plan->n_pos > 0) {
        GEOGCN_REQUIRE(docidx_t && val_t && G, GEOGCN_E_NULL, "xt_dot_f32: null pointer");
        GEOGCN_REQUIRE(ldg % 4 == 0 && ldg >= (int64_t)F4 * 4 && aligned16(G), GEOGCN_E_ALIGN, "xt_dot_f32: bad G pitch / base");
        const size_t need = geogcn_xt_workspace_bytes(plan);
        GEOGCN_REQUIRE(ws && ws_bytes >= need && aligned16(ws), GEOGCN_E_ARG, "xt_dot_f32: workspace too small (%zu < %zu)",
                       ws_bytes, need);
        static const int rendezvous = [] {
            const char* e = getenv("GEOGCN_XT_RENDEZVOUS");       // 0 = free-running sweepers (A/B switch)
            return (e && e[0] == '0') ? 0 : 1;
        }();
        unsigned* arrive = rendezvous ? (unsigned*)ws : nullptr;
        if (arrive) {
            const int zrc = zero_fill_async(arrive, xt_arrive_bytes(plan), st);
            if (zrc) return zrc;
        }
        partial = (float*)((char*)ws + xt_arrive_bytes(plan));
        XtArgs a{docidx_t, val_t, G, ldg, F, plan->d_bin_start, plan->d_bin_words, plan->d_wptr, plan->d_doc_lo,
                 plan->n_words, plan->slots, plan->n_batches, plan->cap, plan->doc_block, partial,
                 (int64_t)plan->K4 * kGroup * 4, plan->n_pos, arrive, plan->max_blocks, 4000};
        const size_t lds = (size_t)plan->cap * plan->K4 * kGroup * sizeof(float4) + (size_t)plan->cap * 2 * sizeof(int);
        const dim3 grid((unsigned)(kNumXCD * plan->slots));
        switch (plan->K4) {
#define GEOGCN_XT(K)                                                                                              \
    case K: {                                                                                                     \
        auto kern = xt_tail_kernel<K>;                                                                            \
        static bool attr_done = false;                                                                            \
        if (!attr_done) {                                                                                         \
            GEOGCN_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
            attr_done = true;                                                                                     \
        }                                                                                                         \
        hipLaunchKernelGGL(kern, grid, dim3(wg_threads(K)), lds, st, a);                                          \
    } break;
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
    hipLaunchKernelGGL(xt_combine_kernel, dim3(cgrid), dim3(kBlock), 0, st, plan->n_words, F, F4, plan->d_pos_of_word,
                       partial, (int64_t)plan->K4 * kGroup * 4, plan->n_pos, dW, ldw);
    GEOGCN_LAUNCH_CHECK("xt_combine_kernel");
    return 0;
}

}  // extern "C"
