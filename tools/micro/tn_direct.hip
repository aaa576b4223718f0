// Micro-benchmark (NOT part of the product): dW[M x N] = H^T . G for the GCN's weight gradients (M = 300, N = 600 -- the highway
// block's dual launch -- K = 440,000 node rows), exact fp32 (v_mfma_f32_16x16x4_f32), WITHOUT LDS and WITHOUT barriers.
// Both operands are k-strided in memory ([K][M], [K][N]) and that is exactly the MFMA's operand layout: lane l of an A (B) fragment
// holds H[k0 + l/16][m0 + l%16] (G[k0 + l/16][n0 + l%16]) -- one buffer_load_dword per fragment, four 64-byte row segments per wave.
// So every wave streams its own fragments straight from L1 / L2 into registers, D steps ahead, and multiplies: 5 + 5 loads and 25 MFMAs
// per 4 k (wave tile 80 x 80); the 8 waves of a block (160 x 320) share lines through the CU's L1 only.  Split-K slabs to a workspace
// (combined by the library's ordered reduce, not timed here: ~10 us).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/tn_direct.hip -o tools/micro/bin/tn_direct && tools/micro/bin/tn_direct
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e_ = (x);                                                   \
        if (e_ != hipSuccess) {                                                \
            printf("%s: %s\n", #x, hipGetErrorString(e_));                     \
            exit(1);                                                           \
        }                                                                      \
    } while (0)

constexpr uint32_t kOob = 0x80000000u;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const float* base, int64_t bytes) {
    const uint32_t n = bytes <= 0 ? 0u : (bytes > 0x7FFFFFFFll ? 0x7FFFFFFFu : (uint32_t)bytes);
    const uint64_t b = reinterpret_cast<uint64_t>(base);
    const uint64_t bu = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(b >> 32)) << 32) |
                        (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)b);
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(bu), 0, __builtin_amdgcn_readfirstlane(n), 0x00020000);
}

// MR x NR tiles of 16 x 16 per wave, WM x WN waves per block, D = k4-steps of loads in flight
template <int MR, int NR, int WM, int WN, int D>
__global__ __launch_bounds__(64 * WM * WN, 1) void tn_direct_kernel(const float* __restrict__ H, int64_t ldh, int M,
                                                                    const float* __restrict__ G, int64_t ldg, int N, int64_t K,
                                                                    float* __restrict__ W, int64_t ldw, int n_mt, int n_nt, int64_t kchunk) {
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wid / WN, wn = wid % WN;
    const int li = lane & 15, lk = lane >> 4;
    const int tile = blockIdx.x % (n_mt * n_nt), z = blockIdx.x / (n_mt * n_nt);
    const int mt = tile / n_nt, nt = tile % n_nt;
    const int m0 = (mt * WM + wm) * MR * 16, n0 = (nt * WN + wn) * NR * 16;
    const int64_t kbeg = (int64_t)z * kchunk, kend = min(K, kbeg + kchunk);
    if (kbeg >= kend) return;
    // descriptors start at the slab's first row: rows past its end read as zeros in hardware
    const __amdgpu_buffer_rsrc_t hr = rsrc(H + kbeg * ldh, (kend - kbeg) * ldh * 4), gr = rsrc(G + kbeg * ldg, (kend - kbeg) * ldg * 4);
    // lane part of the offsets; a fragment whose 16 columns start beyond the matrix (or whose lane's column does) reads zeros
    uint32_t ho[MR], go[NR];
#pragma unroll
    for (int i = 0; i < MR; ++i) ho[i] = (m0 + i * 16 + li < M) ? (uint32_t)((lk * ldh + m0 + i * 16 + li) * 4) : kOob;
#pragma unroll
    for (int j = 0; j < NR; ++j) go[j] = (n0 + j * 16 + li < N) ? (uint32_t)((lk * ldg + n0 + j * 16 + li) * 4) : kOob;
    const uint32_t hstep = (uint32_t)ldh * 16u, gstep = (uint32_t)ldg * 16u;       // 4 rows per k4-step, in bytes
    f32x4 acc[MR][NR];
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nsteps = (int)((kend - kbeg + 3) / 4);
    float ra[D + 1][MR], rb[D + 1][NR];
    auto fetch = [&](float (&a)[MR], float (&b)[NR], int s) {        // (steps past the end: beyond num_records -> zeros)
#pragma unroll
        for (int i = 0; i < MR; ++i) a[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(hr, (int)ho[i], (int)(s * hstep), 0));
#pragma unroll
        for (int j = 0; j < NR; ++j) b[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(gr, (int)go[j], (int)(s * gstep), 0));
    };
#pragma unroll
    for (int d = 0; d < D; ++d) fetch(ra[d], rb[d], d);
#pragma unroll 1
    for (int s0 = 0; s0 < nsteps; s0 += D + 1) {
#pragma unroll
        for (int u = 0; u <= D; ++u) {
            fetch(ra[(u + D) % (D + 1)], rb[(u + D) % (D + 1)], s0 + u + D);
            if (s0 + u < nsteps) {
#pragma unroll
                for (int i = 0; i < MR; ++i)
#pragma unroll
                    for (int j = 0; j < NR; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ra[u][i], rb[u][j], acc[i][j], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // D[i = 4 * (lane / 16) + r][j = lane % 16]  ->  slab z, row m, column n
    float* Wz = W + (int64_t)z * M * ldw;
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + i * 16 + 4 * lk + r, n = n0 + j * 16 + li;
                if (m < M && n < N) Wz[(int64_t)m * ldw + n] = acc[i][j][r];
            }
}

__global__ void reduce_kernel(int M, int N, int nsplit, const float* W, int64_t ldw, float* C) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= M * N) return;
    const int m = e / N, n = e % N;
    float s = 0.f;
    for (int z = 0; z < nsplit; ++z) s += W[((int64_t)z * M + m) * ldw + n];
    C[e] = s;
}

template <int MR, int NR, int WM, int WN, int D>
static void run(const char* name, const float* dH, int M, const float* dG, int N, int64_t K, float* dW, float* dC, const std::vector<float>& hH,
                const std::vector<float>& hG, int nsplit_req) {
    const int BMt = MR * 16 * WM, BNt = NR * 16 * WN;
    const int n_mt = (M + BMt - 1) / BMt, n_nt = (N + BNt - 1) / BNt;
    int nsplit = nsplit_req > 0 ? nsplit_req : 256 / (n_mt * n_nt);
    const int64_t kchunk = ((K + nsplit - 1) / nsplit + 3) / 4 * 4;
    nsplit = (int)((K + kchunk - 1) / kchunk);
    const int grid = n_mt * n_nt * nsplit;
    auto kern = tn_direct_kernel<MR, NR, WM, WN, D>;
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WM * WN), 0, 0, dH, (int64_t)M, M, dG, (int64_t)N, N, K, dW, (int64_t)N, n_mt, n_nt, kchunk);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int reps = 20;
    CK(hipEventRecord(e0));
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WM * WN), 0, 0, dH, (int64_t)M, M, dG, (int64_t)N, N, K, dW, (int64_t)N, n_mt, n_nt, kchunk);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    hipLaunchKernelGGL(reduce_kernel, dim3((M * N + 255) / 256), dim3(256), 0, 0, M, N, nsplit, dW, (int64_t)N, dC);
    std::vector<float> hc((size_t)M * N);
    CK(hipMemcpy(hc.data(), dC, hc.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0;
    const int ms_[] = {0, 1, 15, 16, 79, 80, 159, 160, 299}, ns_[] = {0, 17, 299, 300, 319, 320, 599};
    for (int m : ms_)
        for (int n : ns_) {
            if (m >= M || n >= N) continue;
            double s = 0, sa = 0;
            for (int64_t k = 0; k < K; ++k) {
                const double t = (double)hH[k * M + m] * (double)hG[k * N + n];
                s += t;
                sa += fabs(t);
            }
            worst = fmax(worst, fabs(s - hc[(size_t)m * N + n]) / sa);
        }
    printf("%-44s grid %4d (%d x %d tiles x %d slabs)  %.3f ms  %.1f TF   max err / sum|terms| %.2e\n", name, grid, n_mt, n_nt, nsplit, ms,
           2.0 * M * N * K / ms / 1e9, worst);
}

int main() {
    const int64_t K = 440000;
    const int M = 300;
    for (int N : {600, 300, 256}) {
        std::vector<float> hH((size_t)K * M), hG((size_t)K * N);
        uint32_t s = 12345;
        auto rnd = [&]() {
            s = s * 1664525u + 1013904223u;
            return (float)((s >> 8) & 0xffff) / 65536.f - 0.5f;
        };
        for (auto& x : hH) x = rnd();
        for (auto& x : hG) x = rnd() * 0.1f;
        float *dH, *dG, *dW, *dC;
        CK(hipMalloc(&dH, hH.size() * 4));
        CK(hipMalloc(&dG, hG.size() * 4));
        CK(hipMalloc(&dW, (size_t)256 * M * N * 4));
        CK(hipMalloc(&dC, (size_t)M * N * 4));
        CK(hipMemcpy(dH, hH.data(), hH.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dG, hG.data(), hG.size() * 4, hipMemcpyHostToDevice));
        printf("M = %d, N = %d, K = %lld (%.1f GFLOP)\n", M, N, (long long)K, 2.0 * M * N * K / 1e9);
        run<5, 5, 2, 4, 3>("register-direct, 160 x 320, 3 steps ahead", dH, M, dG, N, K, dW, dC, hH, hG, 0);
        run<5, 5, 2, 4, 5>("register-direct, 160 x 320, 5 steps ahead", dH, M, dG, N, K, dW, dC, hH, hG, 0);
        run<5, 5, 2, 4, 7>("register-direct, 160 x 320, 7 steps ahead", dH, M, dG, N, K, dW, dC, hH, hG, 0);
        run<5, 5, 2, 2, 5>("register-direct, 160 x 160 (4 waves), 5 ahead", dH, M, dG, N, K, dW, dC, hH, hG, 0);
        CK(hipFree(dH)); CK(hipFree(dG)); CK(hipFree(dW)); CK(hipFree(dC));
    }
    return 0;
}
