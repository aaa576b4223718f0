// Micro-benchmark (NOT part of the product): does a per-XCD sliding window over a big table get L2 hits on gfx950?
// Model of the document-blocked X^T.dS0 sweep: 256 workgroups, workgroup b belongs to "virtual XCD" b % 8 which owns
// rows [x*N/8, (x+1)*N/8) of a table of 1280-byte rows and walks that range in windows of `win` rows; inside a window
// every 16-lane group gathers `per_win` random rows of the window (K4 = 5 float4 per lane).  Prints the XCC id the
// first 16 workgroups actually ran on, and the gather bandwidth per window size.  With L2 reuse the bandwidth should
// rise towards the L2-hit rate (~18-30 TB/s) once a window fits the 4 MB L2; without, it stays at the fabric rate (~7).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/xcd_window.hip -o gpurun_out/xcd_window && gpurun_out/xcd_window
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__global__ __launch_bounds__(1024, 1) void window_kernel(const float4* __restrict__ T, int n_rows, int win, int per_win,
                                                         int sync_wg, int xcd_mode, float4* __restrict__ out,
                                                         int* __restrict__ xcc) {
    const int x = xcd_mode ? blockIdx.x % 8 : blockIdx.x / (gridDim.x / 8);
    const int g = threadIdx.x >> 4, l = threadIdx.x & 15;
    if (threadIdx.x == 0 && blockIdx.x < 64) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        xcc[blockIdx.x] = (int)(id & 0xf);
    }
    const int r_lo = (int)((long)n_rows * x / 8), r_hi = (int)((long)n_rows * (x + 1) / 8);
    unsigned s = 12345u + (blockIdx.x * 64 + g) * 2654435761u;
    float4 acc = make_float4(0, 0, 0, 0);
    for (int w0 = r_lo; w0 < r_hi; w0 += win) {
        const int wn = min(win, r_hi - w0);
        for (int it = 0; it < per_win; it += 2) {
            s = s * 1664525u + 1013904223u;
            const int r0 = w0 + (int)(((unsigned long long)(s >> 4) * (unsigned)wn) >> 28);
            s = s * 1664525u + 1013904223u;
            const int r1 = w0 + (int)(((unsigned long long)(s >> 4) * (unsigned)wn) >> 28);
            const float4* p0 = T + (long)r0 * 80 + l;
            const float4* p1 = T + (long)r1 * 80 + l;
            float4 v0[5], v1[5];
#pragma unroll
            for (int k = 0; k < 5; ++k) { v0[k] = p0[16 * k]; v1[k] = p1[16 * k]; }
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                acc.x += v0[k].x + v1[k].x; acc.y += v0[k].y + v1[k].y;
                acc.z += v0[k].z + v1[k].z; acc.w += v0[k].w + v1[k].w;
            }
        }
        if (sync_wg) __syncthreads();
    }
    if (acc.x == 12345.678f) out[blockIdx.x * 64 + g] = acc;
}

int main() {
    const int n_rows = 440000;
    float4* T;
    float4* out;
    int* xcc;
    hipMalloc(&T, (size_t)n_rows * 1280);
    hipMemset(T, 0, (size_t)n_rows * 1280);
    hipMalloc(&out, 256 * 64 * sizeof(float4));
    hipMalloc(&xcc, 64 * sizeof(int));
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    // total gathers per group fixed: 14.6 GB in all  (11.4 M rows of 1280 B / (256 * 64 groups) = 696 per group)
    const int per_group = 696;
    for (int xcd_mode = 1; xcd_mode >= 0; --xcd_mode)
        for (int sync_wg = 0; sync_wg <= 1; ++sync_wg)
            for (int win : {256, 512, 1024, 2048, 4096, 55000}) {
                const int n_win = (n_rows / 8 + win - 1) / win;
                int per_win = (per_group + n_win - 1) / n_win;
                per_win += per_win & 1;
                for (int rep = 0; rep < 2; ++rep) {
                    hipEventRecord(a);
                    hipLaunchKernelGGL(window_kernel, dim3(256), dim3(1024), 0, 0, T, n_rows, win, per_win, sync_wg, xcd_mode, out,
                                       xcc);
                    hipEventRecord(b);
                    hipEventSynchronize(b);
                }
                float ms;
                hipEventElapsedTime(&ms, a, b);
                const double bytes = 256.0 * 64 * (double)per_win * n_win * 1280;
                printf("map %s  wg-sync %d  window %5d rows (%.2f MB)  %3d gathers/group/window: %.3f ms  %.1f TB/s\n",
                       xcd_mode ? "b%8 " : "b/32", sync_wg, win, win * 1280 / 1e6, per_win, ms, bytes / ms / 1e9);
            }
    int h[64];
    hipMemcpy(h, xcc, sizeof(h), hipMemcpyDeviceToHost);
    printf("XCC id of workgroups 0..31:");
    for (int i = 0; i < 32; ++i) printf(" %d", h[i]);
    printf("\n");
    return 0;
}
