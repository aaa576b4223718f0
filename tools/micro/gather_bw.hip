// Micro-benchmark (NOT part of the product): random row-gather bandwidth vs working-set size on gfx950.
// Each 16-lane group reads `row_bytes` contiguous bytes (float4 per lane, K4 per lane) from a random row
// of a table of `n_rows` rows; rows drawn uniformly (LCG) -> measures L2 / MALL / HBM gather throughput.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/gather_bw.hip -o gpurun_out/gather_bw && ./gather_bw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef float f4v __attribute__((ext_vector_type(4)));
template <int NT>
__device__ __forceinline__ float4 ld4(const float4* p) {
    if (NT) { f4v v = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(p)); return make_float4(v.x, v.y, v.z, v.w); }
    return *p;
}
template <int K4, int NT>
__global__ __launch_bounds__(256) void gather_kernel(const float4* __restrict__ T, long row_f4, unsigned n_rows,
                                                     int iters, float4* __restrict__ out, unsigned seed) {
    const int g = (blockIdx.x * 256 + threadIdx.x) >> 4;
    const int l = threadIdx.x & 15;
    unsigned s = seed + g * 2654435761u;
    float4 acc = make_float4(0, 0, 0, 0);
    for (int it = 0; it < iters; it += 2) {
        s = s * 1664525u + 1013904223u;
        const unsigned r0 = (unsigned)(((unsigned long long)(s >> 4) * n_rows) >> 28);
        s = s * 1664525u + 1013904223u;
        const unsigned r1 = (unsigned)(((unsigned long long)(s >> 4) * n_rows) >> 28);
        const float4* p0 = T + (long)r0 * row_f4 + l;
        const float4* p1 = T + (long)r1 * row_f4 + l;
        float4 v0[K4], v1[K4];
#pragma unroll
        for (int k = 0; k < K4; ++k) { v0[k] = ld4<NT>(p0 + 16 * k); v1[k] = ld4<NT>(p1 + 16 * k); }
#pragma unroll
        for (int k = 0; k < K4; ++k) {
            acc.x += v0[k].x + v1[k].x; acc.y += v0[k].y + v1[k].y;
            acc.z += v0[k].z + v1[k].z; acc.w += v0[k].w + v1[k].w;
        }
    }
    if (acc.x == 12345.678f) out[g] = acc;
}

template <int K4, int NT>
double run(const float4* T, long row_f4, unsigned n_rows, int groups, int iters, float4* out) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int blocks = groups / 16;
    gather_kernel<K4, NT><<<blocks, 256>>>(T, row_f4, n_rows, iters, out, 1u);
    hipDeviceSynchronize();
    hipEventRecord(a);
    gather_kernel<K4, NT><<<blocks, 256>>>(T, row_f4, n_rows, iters, out, 7u);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double bytes = (double)groups * iters * K4 * 256.0;
    return bytes / ms / 1e9;   // TB/s
}

int main(int argc, char** argv) {
    if (argc >= 4 && !strcmp(argv[1], "--json")) {
        // one point, machine-readable (bench.py: roofline.gather_ceiling): `--json <table MB> <row bytes: 256 | 512 | 1280>`, best of 3
        const size_t bytes = (size_t)atol(argv[2]) << 20;
        const int row = atoi(argv[3]);
        float4* T; float4* out;
        if (hipMalloc(&T, bytes) != hipSuccess || hipMalloc(&out, 1 << 24) != hipSuccess) { printf("{\"error\": \"hipMalloc\"}\n"); return 1; }
        hipMemset(T, 0, bytes);
        const int groups = 256 * 8 * 16 * 4;
        double best = 0;
        for (int rep = 0; rep < 3; ++rep) {
            const double r = row == 256 ? run<1, 0>(T, 16, (unsigned)(bytes / 256), groups, 64, out)
                           : row == 512 ? run<2, 0>(T, 32, (unsigned)(bytes / 512), groups, 32, out)
                                        : run<5, 0>(T, 80, (unsigned)(bytes / 1280), groups, 16, out);
            if (r > best) best = r;
        }
        printf("{\"table_mb\": %zu, \"row_bytes\": %d, \"tbps\": %.3f}\n", bytes >> 20, row == 256 || row == 512 ? row : 1280, best);
        return 0;
    }
    const size_t maxb = (size_t)4 << 30;
    float4* T; float4* out;
    hipMalloc(&T, maxb); hipMalloc(&out, 1 << 24);
    hipMemset(T, 0, maxb);
    const int groups = 256 * 8 * 16 * 4;      // 8 blocks/CU x 16 groups x 4 rounds
    printf("%10s %8s | TB/s for row bytes: 256 (K4=1)   512 (K4=2)   1280 (K4=5)\n", "table", "rows@256");
    for (size_t mb : {1, 2, 3, 4, 8, 16, 32, 64, 128, 192, 256, 384, 512, 1024, 2048, 4096}) {
        const size_t bytes = mb << 20;
        double r[5];
        r[0] = run<1, 0>(T, 16, (unsigned)(bytes / 256), groups, 64, out);
        r[1] = run<2, 0>(T, 32, (unsigned)(bytes / 512), groups, 32, out);
        r[2] = run<5, 0>(T, 80, (unsigned)(bytes / 1280), groups, 16, out);
        r[3] = run<1, 1>(T, 16, (unsigned)(bytes / 256), groups, 64, out);
        r[4] = run<5, 1>(T, 80, (unsigned)(bytes / 1280), groups, 16, out);
        printf("%8zu MB %8zu | %28.2f %12.2f %12.2f   nt: %6.2f %6.2f\n", mb, bytes / 256, r[0], r[1], r[2], r[3], r[4]);
    }
    return 0;
}
