// Micro-benchmark (NOT part of the product): does a non-temporal hint on the COLD gathers keep the HOT
// rows resident in the 4 MB per-XCD L2?  Each 8-lane group alternates one gather from a small hot table
// (hot_mb MB, 128-byte rows) with `cold_per_hot` gathers from a 2 GB cold table; cold loads are issued
// plain / nontemporal.  Reports total gather TB/s; higher with nt => hot rows survive.
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f4v __attribute__((ext_vector_type(4)));
template <int NT>
__device__ __forceinline__ float4 ld(const float4* p) {
    if (NT) {
        f4v v = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(p));
        return make_float4(v.x, v.y, v.z, v.w);
    }
    return *p;
}

template <int NT>
__global__ __launch_bounds__(256) void k(const float4* __restrict__ hot, unsigned hot_rows, const float4* __restrict__ cold,
                                         unsigned cold_rows, int iters, int cold_per_hot, float4* out, unsigned seed) {
    const int g = (blockIdx.x * 256 + threadIdx.x) >> 3;
    const int l = threadIdx.x & 7;
    unsigned s = seed + g * 2654435761u;
    float4 acc = make_float4(0, 0, 0, 0);
    for (int it = 0; it < iters; ++it) {
        s = s * 1664525u + 1013904223u;
        const unsigned rh = (unsigned)(((unsigned long long)(s >> 4) * hot_rows) >> 28);
        float4 v = hot[(long)rh * 8 + l];
        acc.x += v.x; acc.y += v.y;
        for (int c = 0; c < cold_per_hot; ++c) {
            s = s * 1664525u + 1013904223u;
            const unsigned rc = (unsigned)(((unsigned long long)(s >> 4) * cold_rows) >> 28);
            float4 w = ld<NT>(cold + (long)rc * 8 + l);
            acc.z += w.x; acc.w += w.y;
        }
    }
    if (acc.x == 12345.678f) out[g] = acc;
}

int main() {
    float4 *hot, *cold, *out;
    const size_t cold_b = (size_t)2 << 30;
    hipMalloc(&hot, 64 << 20); hipMalloc(&cold, cold_b); hipMalloc(&out, 64 << 20);
    hipMemset(hot, 0, 64 << 20); hipMemset(cold, 0, cold_b);
    const int groups = 256 * 8 * 32 * 4;
    printf("hot MB  cold/hot   plain TB/s   nt TB/s\n");
    for (int hot_mb : {1, 2, 3}) for (int cph : {1, 2, 4}) {
        double r[2];
        for (int nt = 0; nt < 2; ++nt) {
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            const unsigned hr = (unsigned)((size_t)hot_mb << 20) / 128, cr = (unsigned)(cold_b / 128);
            const int iters = 64;
            for (int rep = 0; rep < 2; ++rep) {
                if (rep == 1) hipEventRecord(a);
                if (nt) k<1><<<groups / 32, 256>>>(hot, hr, cold, cr, iters, cph, out, 7u + rep);
                else k<0><<<groups / 32, 256>>>(hot, hr, cold, cr, iters, cph, out, 7u + rep);
            }
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            r[nt] = (double)groups * iters * (1 + cph) * 128.0 / ms / 1e9;
        }
        printf("%6d %9d %12.2f %9.2f\n", hot_mb, cph, r[0], r[1]);
    }
    return 0;
}
