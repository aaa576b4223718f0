// Micro-benchmark (NOT part of the product): A^T . B (the weight gradients dW = H^T . dZ, reduction over the 440,000 nodes) with the
// fp32-class split-bf16 ("bf16x3") contraction on v_mfma_f32_16x16x32_bf16.
// Both operands are k-STRIDED fp32 in memory ([K][M], [K][N]).  A thread loads an 8 (k) x 4 (columns) patch as eight float4s, splits
// every value EXACTLY into three bf16 terms and writes, per column, three 16-byte k-contiguous pieces (one per plane): the transpose
// happens in registers and the LDS images are [plane][row][32 k] -- the fragment layout of the MFMA.  One 8-wave block per CU, tile
// 160 x 320 (2 x 4 waves, wave tile 80 x 80), ONE LDS image of three planes (115 KB: no room for a second), split-K slabs in fp32.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/x3_tn.hip -o tools/micro/bin/x3_tn && tools/micro/bin/x3_tn
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e_ = (x);                                                   \
        if (e_ != hipSuccess) {                                                \
            printf("%s: %s\n", #x, hipGetErrorString(e_));                     \
            exit(1);                                                           \
        }                                                                      \
    } while (0)

constexpr int kNumXCD = 8;
constexpr int BKH = 32, ROWB = 80;

__device__ __forceinline__ uint32_t bf16_pack(float lo, float hi) {
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ void split3_pair(float x0, float x1, uint32_t (&p)[3]) {
    p[0] = bf16_pack(x0, x1);
    const float r0 = x0 - __uint_as_float(p[0] << 16), r1 = x1 - __uint_as_float(p[0] & 0xffff0000u);
    p[1] = bf16_pack(r0, r1);
    const float s0 = r0 - __uint_as_float(p[1] << 16), s1 = r1 - __uint_as_float(p[1] & 0xffff0000u);
    p[2] = bf16_pack(s0, s1);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t mk_rsrc(const void* base, int64_t bytes) {
    const uint32_t n = bytes <= 0 ? 0u : (bytes > 0x7FFFFFFFll ? 0x7FFFFFFFu : (uint32_t)bytes);
    const uint64_t b = reinterpret_cast<uint64_t>(base);
    const uint64_t bu = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(b >> 32)) << 32) |
                        (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)b);
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(bu), 0, __builtin_amdgcn_readfirstlane(n), 0x00020000);
}

// PF = stages of global loads in flight per thread.  PROBE: 1 = no global loads, 2 = no LDS stores (and no split), 4 = no MFMAs
template <int BM, int BN, int PF, int PROBE = 0, int IG = 0>
__global__ __launch_bounds__(512, 1) void x3_tn_kernel(const float* __restrict__ A, int64_t lda, int64_t M, const float* __restrict__ B,
                                                       int64_t ldb, int64_t N, int64_t K, float* __restrict__ W, int64_t ldw, int n_mt,
                                                       int n_nt, int nsplit, int64_t kchunk) {
    constexpr int NTH = 512, kASplit = 192;
    constexpr int MR = BM / 32, NR = BN / 64;
    constexpr int kAItems = 4 * (BM / 4), kBItems = 4 * (BN / 4);
    static_assert(kAItems <= kASplit && kBItems <= NTH - kASplit, "patch lists must fit the thread ranges");
    constexpr int kImgA = BM * ROWB, kPlane = (BM + BN) * ROWB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 2, wn = wid & 3, li = lane & 15, lg = lane >> 4;

    const int b = blockIdx.x;
    const int xcd = b % kNumXCD, s = b / kNumXCD;
    const int tiles = n_nt * n_mt;
    const int tile = s % tiles;
    const int nt = __builtin_amdgcn_readfirstlane(tile % n_nt);
    const int mt = __builtin_amdgcn_readfirstlane(tile / n_nt);
    const int z = __builtin_amdgcn_readfirstlane(xcd + kNumXCD * (s / tiles));
    if (z >= nsplit) return;
    const int64_t m0 = (int64_t)mt * BM, n0 = (int64_t)nt * BN;
    const int64_t kbeg = (int64_t)z * kchunk, kend = K < kbeg + kchunk ? K : kbeg + kchunk;
    const int nk = (int)((kend - kbeg + BKH - 1) / BKH);

    const bool isA = tid < kASplit;
    const int it = isA ? tid : tid - kASplit;
    const int cols4 = isA ? BM / 4 : BN / 4;
    const bool active = it < 4 * cols4;
    const int k8 = it / cols4, c4 = it % cols4;
    const float* P = isA ? A : B;
    const int64_t ld = isA ? lda : ldb;
    const int64_t c0 = isA ? m0 : n0;
    const int64_t ctot = isA ? M : N;
    const bool col_ok = active && (c4 * 4 < ((ctot + 3) & ~(int64_t)3) - c0);
    f32x4 ring[PF][8];
    auto gload = [&](f32x4 (&r)[8], int kt) {
        const int64_t k0 = kbeg + (int64_t)kt * BKH;
        const __amdgpu_buffer_rsrc_t rs = mk_rsrc(P + k0 * ld + c0, ((kend - k0) * ld - c0) * 4);
        const uint32_t ld4 = (uint32_t)ld * 4u;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint32_t off = (uint32_t)(k8 * 8 + i) * ld4 + (uint32_t)c4 * 16u;
            r[i] = (PROBE & 1) ? f32x4{1.f, 2.f, 3.f, 4.f} : __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(col_ok ? off : 0x80000000u), 0, 0));
        }
    };
    // registers -> the three planes of the LDS image: column e of the patch = component e of every row
    auto sstore = [&](const f32x4 (&r)[8]) {
        if (!active || (PROBE & 2)) return;
        unsigned char* img = smem_raw + (isA ? 0 : kImgA);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            uint32_t p[4][3];
#pragma unroll
            for (int q = 0; q < 4; ++q) split3_pair(r[2 * q][e], r[2 * q + 1][e], p[q]);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                *reinterpret_cast<uint4*>(img + pl * kPlane + (c4 * 4 + e) * ROWB + k8 * 16) = make_uint4(p[0][pl], p[1][pl], p[2][pl], p[3][pl]);
        }
    };

    f32x4 acc[MR][NR];
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int d = 0; d < PF; ++d) gload(ring[d], d);
#pragma unroll 1
    for (int kt0 = 0; kt0 < nk; kt0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int kt = kt0 + u;
            const bool live = kt < nk;          // (no `break`: the ring's slots must stay compile-time constants)
            __syncthreads();                    // everybody done reading the stage before
            if (live) sstore(ring[u]);
            gload(ring[u], kt + PF);            // (past the end of the slab: beyond num_records -> zeros, never multiplied)
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
            if (live && !(PROBE & 4)) {
                const unsigned char* As = smem_raw;
                const unsigned char* Bs = smem_raw + kImgA;
                constexpr int GI = IG ? IG : MR;          // row tiles whose A fragments are held at a time
#pragma unroll
                for (int i0 = 0; i0 < MR; i0 += GI) {
                    bf16x8 af[GI][3];
#pragma unroll
                    for (int i = 0; i < GI; ++i)
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl)
                            if (i0 + i < MR) af[i][pl] = *reinterpret_cast<const bf16x8*>(As + pl * kPlane + (wm * (BM / 2) + (i0 + i) * 16 + li) * ROWB + lg * 16);
#pragma unroll
                    for (int j = 0; j < NR; ++j) {
                        bf16x8 bf[3];
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl)
                            bf[pl] = *reinterpret_cast<const bf16x8*>(Bs + pl * kPlane + (wn * (BN / 4) + j * 16 + li) * ROWB + lg * 16);
#define X3_TERM(PB, PA)                                                                                         \
    _Pragma("unroll") for (int i = 0; i < GI; ++i) if (i0 + i < MR) acc[i0 + i][j] =                            \
        __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[PB], af[i][PA], acc[i0 + i][j], 0, 0, 0);
                        X3_TERM(0, 2) X3_TERM(2, 0) X3_TERM(1, 1) X3_TERM(0, 1) X3_TERM(1, 0) X3_TERM(0, 0)
#undef X3_TERM
                    }
                }
#if 0
#undef X3_TERM
                }
#endif
            }
        }
    }
    float* Wz = W + (int64_t)z * M * ldw;
    const int64_t n_store = (N + 3) & ~(int64_t)3;
#pragma unroll
    for (int i = 0; i < MR; ++i) {
        const int64_t row = m0 + wm * (BM / 2) + i * 16 + li;
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const int64_t col0 = n0 + wn * (BN / 4) + j * 16 + lg * 4;
            f32x4 x = acc[i][j];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (col0 + q >= N) x[q] = 0.f;
            if (row < M && col0 < n_store) *reinterpret_cast<f32x4*>(Wz + row * ldw + col0) = x;
        }
    }
}


// ---- v2: 16-k stages (v_mfma_f32_16x16x16_bf16), TWO three-plane LDS images (2 x 69 KB): the split + store of stage s + 1 runs
// under the MFMAs of stage s; one barrier per stage.  The two 4-wave halves of the block take the loading of alternate stages, so that
// on every SIMD one wave splits while the other multiplies.
typedef short bf16x4 __attribute__((ext_vector_type(4)));
constexpr int BK2 = 16, ROWB2 = 48;          // 16 bf16 = 32 B + 16 B pad: conflict-free ds_read_b64 (12 dwords per row)

template <int BM, int BN, int PF, int PROBE = 0>
__global__ __launch_bounds__(512, 1) void x3_tn2_kernel(const float* __restrict__ A, int64_t lda, int64_t M, const float* __restrict__ B,
                                                        int64_t ldb, int64_t N, int64_t K, float* __restrict__ W, int64_t ldw, int n_mt,
                                                        int n_nt, int nsplit, int64_t kchunk) {
    constexpr int MR = BM / 32, NR = BN / 64;
    constexpr int kAItems = 2 * (BM / 4), kBItems = 2 * (BN / 4);          // patches of 8 k x 4 columns per 16-k stage
    static_assert(kAItems + kBItems <= 256, "a stage's patches must fit one half of the block");
    constexpr int kImgA = BM * ROWB2, kPlane = (BM + BN) * ROWB2, kBuf = 3 * kPlane;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 2, wn = wid & 3, li = lane & 15, lg = lane >> 4;
    const int grp = wid >> 2;                   // which stages this half of the block loads: kt % 2 == grp
    const int b = blockIdx.x;
    const int xcd = b % kNumXCD, s = b / kNumXCD;
    const int tiles = n_nt * n_mt;
    const int tile = s % tiles;
    const int nt = __builtin_amdgcn_readfirstlane(tile % n_nt);
    const int mt = __builtin_amdgcn_readfirstlane(tile / n_nt);
    const int z = __builtin_amdgcn_readfirstlane(xcd + kNumXCD * (s / tiles));
    if (z >= nsplit) return;
    const int64_t m0 = (int64_t)mt * BM, n0 = (int64_t)nt * BN;
    const int64_t kbeg = (int64_t)z * kchunk, kend = K < kbeg + kchunk ? K : kbeg + kchunk;
    const int nk = (int)((kend - kbeg + BK2 - 1) / BK2);

    const int t2 = tid & 255;
    const bool isA = t2 < kAItems;
    const int it = isA ? t2 : t2 - kAItems;
    const int cols4 = isA ? BM / 4 : BN / 4;
    const bool active = it < 2 * cols4;
    const int k8 = it / cols4, c4 = it % cols4;
    const float* P = isA ? A : B;
    const int64_t ld = isA ? lda : ldb;
    const int64_t c0 = isA ? m0 : n0;
    const int64_t ctot = isA ? M : N;
    const bool col_ok = active && (c4 * 4 < ((ctot + 3) & ~(int64_t)3) - c0);
    f32x4 ring[PF][8];
    auto gload = [&](f32x4 (&r)[8], int kt) {
        const int64_t k0 = kbeg + (int64_t)kt * BK2;
        const __amdgpu_buffer_rsrc_t rs = mk_rsrc(P + k0 * ld + c0, ((kend - k0) * ld - c0) * 4);
        const uint32_t ld4 = (uint32_t)ld * 4u;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint32_t off = (uint32_t)(k8 * 8 + i) * ld4 + (uint32_t)c4 * 16u;
            r[i] = (PROBE & 1) ? f32x4{1.f, 2.f, 3.f, 4.f} : __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(col_ok ? off : 0x80000000u), 0, 0));
        }
    };
    auto sstore = [&](int buf, const f32x4 (&r)[8]) {
        if (!active || (PROBE & 2)) return;
        unsigned char* img = smem_raw + buf * kBuf + (isA ? 0 : kImgA);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            uint32_t p[4][3];
#pragma unroll
            for (int q = 0; q < 4; ++q) split3_pair(r[2 * q][e], r[2 * q + 1][e], p[q]);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                *reinterpret_cast<uint4*>(img + pl * kPlane + (c4 * 4 + e) * ROWB2 + k8 * 16) = make_uint4(p[0][pl], p[1][pl], p[2][pl], p[3][pl]);
        }
    };
    f32x4 acc[MR][NR];
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // my stages: grp, grp + 2, grp + 4, ...; ring slot d holds my (d)-th next stage
#pragma unroll
    for (int d = 0; d < PF; ++d) gload(ring[d], grp + 2 * d);
    if (grp == 0) sstore(0, ring[0]);            // stage 0 -> image 0
    if (grp == 0) gload(ring[0], 2 * PF);
    __syncthreads();
    // (stage kt is multiplied from image kt % 2; before that its loader half writes stage kt + 1 into the other image)
    // ring bookkeeping: group 0 has consumed slot 0 (stage 0) above; its next is stage 2 = slot 1 ... handled by a running slot index in
    // an unrolled-by-(2 PF) loop so that slots stay compile-time constants
#pragma unroll 1
    for (int kt0 = 0; kt0 < nk; kt0 += 2 * PF) {
#pragma unroll
        for (int u = 0; u < 2 * PF; ++u) {
            const int kt = kt0 + u;
            const bool live = kt < nk;
            // the loader of stage kt + 1 is group (kt + 1) % 2 = (u + 1) % 2; for that group stage kt + 1 sits in ring slot:
            //   group 0 loads even stages: stage 2 q in slot q % PF (slot 0 was stage 0); group 1 loads odd stages: stage 2 q + 1 in slot q % PF
            constexpr int dummy = 0;
            (void)dummy;
            if (((u + 1) & 1) == grp) {
                const int q = (kt + 1) >> 1;              // my q-th stage
                const int slot_c = ((u + 1) >> 1) % PF;   // compile-time: kt0 is a multiple of 2 PF
                (void)q;
                if (kt + 1 < nk) sstore((kt + 1) & 1, ring[slot_c]);
                gload(ring[slot_c], kt + 1 + 2 * PF);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (live && !(PROBE & 4)) {
                const unsigned char* As = smem_raw + (kt & 1) * kBuf;
                const unsigned char* Bs = As + kImgA;
                bf16x4 af[MR][3];
#pragma unroll
                for (int i = 0; i < MR; ++i)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
                        af[i][pl] = *reinterpret_cast<const bf16x4*>(As + pl * kPlane + (wm * (BM / 2) + i * 16 + li) * ROWB2 + lg * 8);
#pragma unroll
                for (int j = 0; j < NR; ++j) {
                    bf16x4 bf[3];
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
                        bf[pl] = *reinterpret_cast<const bf16x4*>(Bs + pl * kPlane + (wn * (BN / 4) + j * 16 + li) * ROWB2 + lg * 8);
#define X3_TERM2(PB, PA)                                                                                         \
    _Pragma("unroll") for (int i = 0; i < MR; ++i) acc[i][j] =                                                   \
        __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(bf[PB], af[i][PA], acc[i][j], 0, 0, 0);
                    X3_TERM2(0, 2) X3_TERM2(2, 0) X3_TERM2(1, 1) X3_TERM2(0, 1) X3_TERM2(1, 0) X3_TERM2(0, 0)
#undef X3_TERM2
                }
            }
            __syncthreads();
        }
    }
    float* Wz = W + (int64_t)z * M * ldw;
    const int64_t n_store = (N + 3) & ~(int64_t)3;
#pragma unroll
    for (int i = 0; i < MR; ++i) {
        const int64_t row = m0 + wm * (BM / 2) + i * 16 + li;
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const int64_t col0 = n0 + wn * (BN / 4) + j * 16 + lg * 4;
            f32x4 x = acc[i][j];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (col0 + q >= N) x[q] = 0.f;
            if (row < M && col0 < n_store) *reinterpret_cast<f32x4*>(Wz + row * ldw + col0) = x;
        }
    }
}


// ---- v3 (round 6): ONE image, the split moved out from between the barriers.  The exact three-way split of stage kt + 1 is formed IN
// REGISTERS (12 x uint4 per thread) during the MFMA phase of stage kt; between the two barriers of a stage only the 12 ds_write_b128
// remain.  The two 4-wave halves of the block run the phase in opposite order (half 0: split, then multiply; half 1: multiply, then
// split), so that on every SIMD one wave feeds the VALU while the other feeds the matrix pipe.  Global loads are two stages ahead
// (the patch of stage kt + 2 is requested as soon as stage kt + 1 has been split).
template <int BM, int BN, int IG, int PROBE = 0, int ORDER = 0>
__global__ __launch_bounds__(512, 1) void x3_tn3_kernel(const float* __restrict__ A, int64_t lda, int64_t M, const float* __restrict__ B,
                                                        int64_t ldb, int64_t N, int64_t K, float* __restrict__ W, int64_t ldw, int n_mt,
                                                        int n_nt, int nsplit, int64_t kchunk) {
    constexpr int NTH = 512, kASplit = 192;
    constexpr int MR = BM / 32, NR = BN / 64;
    constexpr int kAItems = 4 * (BM / 4), kBItems = 4 * (BN / 4);
    static_assert(kAItems <= kASplit && kBItems <= NTH - kASplit, "patch lists must fit the thread ranges");
    constexpr int kImgA = BM * ROWB, kPlane = (BM + BN) * ROWB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 2, wn = wid & 3, li = lane & 15, lg = lane >> 4;
    const int b = blockIdx.x;
    const int xcd = b % kNumXCD, s = b / kNumXCD;
    const int tiles = n_nt * n_mt;
    const int tile = s % tiles;
    const int nt = __builtin_amdgcn_readfirstlane(tile % n_nt);
    const int mt = __builtin_amdgcn_readfirstlane(tile / n_nt);
    const int z = __builtin_amdgcn_readfirstlane(xcd + kNumXCD * (s / tiles));
    if (z >= nsplit) return;
    const int64_t m0 = (int64_t)mt * BM, n0 = (int64_t)nt * BN;
    const int64_t kbeg = (int64_t)z * kchunk, kend = K < kbeg + kchunk ? K : kbeg + kchunk;
    const int nk = (int)((kend - kbeg + BKH - 1) / BKH);

    // patches are dealt so that BOTH halves of the block (waves 0-3, 4-7) hold A and B patches alike: thread t of a half takes item 2 t' + half
    const bool isA = tid < kASplit;
    const int it = isA ? tid : tid - kASplit;
    const int cols4 = isA ? BM / 4 : BN / 4;
    const bool active = it < 4 * cols4;
    const int k8 = it / cols4, c4 = it % cols4;
    const float* P = isA ? A : B;
    const int64_t ld = isA ? lda : ldb;
    const int64_t c0 = isA ? m0 : n0;
    const int64_t ctot = isA ? M : N;
    const bool col_ok = active && (c4 * 4 < ((ctot + 3) & ~(int64_t)3) - c0);
    f32x4 patch[8];
    uint4 S[4][3];
    auto gload = [&](int kt) {
        const int64_t k0 = kbeg + (int64_t)kt * BKH;
        const __amdgpu_buffer_rsrc_t rs = mk_rsrc(P + k0 * ld + c0, ((kend - k0) * ld - c0) * 4);
        const uint32_t ld4 = (uint32_t)ld * 4u;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint32_t off = (uint32_t)(k8 * 8 + i) * ld4 + (uint32_t)c4 * 16u;
            patch[i] = (PROBE & 1) ? f32x4{1.f, 2.f, 3.f, 4.f} : __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(col_ok ? off : 0x80000000u), 0, 0));
        }
    };
    auto split = [&]() {
        if (PROBE & 2) return;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            uint32_t p[4][3];
#pragma unroll
            for (int q = 0; q < 4; ++q) split3_pair(patch[2 * q][e], patch[2 * q + 1][e], p[q]);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) S[e][pl] = make_uint4(p[0][pl], p[1][pl], p[2][pl], p[3][pl]);
        }
    };
    auto swrite = [&]() {
        if (!active || (PROBE & 2)) return;
        unsigned char* img = smem_raw + (isA ? 0 : kImgA);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<uint4*>(img + pl * kPlane + (c4 * 4 + e) * ROWB + k8 * 16) = S[e][pl];
    };
    f32x4 acc[MR][NR];
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto multiply = [&]() {
        if (PROBE & 4) return;
        const unsigned char* As = smem_raw;
        const unsigned char* Bs = smem_raw + kImgA;
        constexpr int GI = IG ? IG : MR;
#pragma unroll
        for (int i0 = 0; i0 < MR; i0 += GI) {
            bf16x8 af[GI][3];
#pragma unroll
            for (int i = 0; i < GI; ++i)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    if (i0 + i < MR) af[i][pl] = *reinterpret_cast<const bf16x8*>(As + pl * kPlane + (wm * (BM / 2) + (i0 + i) * 16 + li) * ROWB + lg * 16);
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                bf16x8 bf[3];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    bf[pl] = *reinterpret_cast<const bf16x8*>(Bs + pl * kPlane + (wn * (BN / 4) + j * 16 + li) * ROWB + lg * 16);
#define X3_TERM(PB, PA)                                                                                         \
    _Pragma("unroll") for (int i = 0; i < GI; ++i) if (i0 + i < MR) acc[i0 + i][j] =                            \
        __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[PB], af[i][PA], acc[i0 + i][j], 0, 0, 0);
                X3_TERM(0, 2) X3_TERM(2, 0) X3_TERM(1, 1) X3_TERM(0, 1) X3_TERM(1, 0) X3_TERM(0, 0)
#undef X3_TERM
            }
        }
    };
    const bool first = ORDER == 0 ? (wid < 4) : ORDER == 1 ? true : false;      // split before multiplying?  (ORDER 1 / 2: every wave the same way)
    gload(0);
    split();
    gload(1);
#pragma unroll 1
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();                    // everybody done reading the stage before
        swrite();
        __syncthreads();
        if (first) {
            split();                        // stage kt + 1 (requested a whole stage ago)
            gload(kt + 2);
            __builtin_amdgcn_sched_barrier(0);
            multiply();
        } else {
            multiply();
            __builtin_amdgcn_sched_barrier(0);
            split();
            gload(kt + 2);
        }
    }
    float* Wz = W + (int64_t)z * M * ldw;
    const int64_t n_store = (N + 3) & ~(int64_t)3;
#pragma unroll
    for (int i = 0; i < MR; ++i) {
        const int64_t row = m0 + wm * (BM / 2) + i * 16 + li;
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const int64_t col0 = n0 + wn * (BN / 4) + j * 16 + lg * 4;
            f32x4 x = acc[i][j];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (col0 + q >= N) x[q] = 0.f;
            if (row < M && col0 < n_store) *reinterpret_cast<f32x4*>(Wz + row * ldw + col0) = x;
        }
    }
}


// ---- v4 (round 6): TWO 4-wave blocks per CU, tile 160 x 160 each (one three-plane image of 76.8 KB per block): the blocks drift apart and
// one multiplies while the other splits.  Price: 8 k x 4 column patches per flop up by a third (320 patches for 160 x 160 against 480
// for 160 x 320).  The 320 patches of a stage go to 256 threads: thread t takes patch t and, if t < 64, patch t + 256 (BAL = 0), or
// -- BAL = 1 -- wave w takes the k8 group w and lane l the columns l + 64 q, q = 0..4, as 8 k x 1 column items (five per thread).
template <int PROBE = 0, int BAL = 0>
__global__ __launch_bounds__(256, 2) void x3_tn4_kernel(const float* __restrict__ A, int64_t lda, int64_t M, const float* __restrict__ B,
                                                        int64_t ldb, int64_t N, int64_t K, float* __restrict__ W, int64_t ldw, int n_mt,
                                                        int n_nt, int nsplit, int64_t kchunk) {
    constexpr int BM = 160, BN = 160, NTH = 256;
    constexpr int MR = BM / 32, NR = BN / 32;                // 2 x 2 waves, wave tile 80 x 80
    constexpr int kImgA = BM * ROWB, kPlane = (BM + BN) * ROWB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wn = wid & 1, li = lane & 15, lg = lane >> 4;
    const int b = blockIdx.x;
    const int xcd = b % kNumXCD, s = b / kNumXCD;
    const int tiles = n_nt * n_mt;
    const int tile = s % tiles;
    const int nt = __builtin_amdgcn_readfirstlane(tile % n_nt);
    const int mt = __builtin_amdgcn_readfirstlane(tile / n_nt);
    const int z = __builtin_amdgcn_readfirstlane(xcd + kNumXCD * (s / tiles));
    if (z >= nsplit) return;
    const int64_t m0 = (int64_t)mt * BM, n0 = (int64_t)nt * BN;
    const int64_t kbeg = (int64_t)z * kchunk, kend = K < kbeg + kchunk ? K : kbeg + kchunk;
    const int nk = (int)((kend - kbeg + BKH - 1) / BKH);

    // patch lists: items [0, 160) = A (4 k8 x 40 c4), [160, 320) = B
    struct Item { bool isA, ok; int k8, c4; };
    auto item_of = [&](int it) {
        Item x;
        x.isA = it < 4 * (BM / 4);
        const int q = x.isA ? it : it - 4 * (BM / 4);
        const int cols4 = x.isA ? BM / 4 : BN / 4;
        x.k8 = q / cols4; x.c4 = q % cols4;
        const int64_t ctot = x.isA ? M : N, c0 = x.isA ? m0 : n0;
        x.ok = x.c4 * 4 < ((ctot + 3) & ~(int64_t)3) - c0;
        return x;
    };
    const Item it0 = item_of(tid), it1 = item_of(tid + NTH);
    const bool two = tid < 4 * (BM / 4) + 4 * (BN / 4) - NTH;       // the first 64 threads (wave 0) take a second patch
    f32x4 p0[8], p1[8];
    // (A / B differs per lane only in wave 2: there both descriptors are read, each lane out of bounds on the one that is not its own)
    auto gload1 = [&](f32x4 (&r)[8], const Item& x, int kt, bool on) {
        const int64_t k0 = kbeg + (int64_t)kt * BKH;
        const __amdgpu_buffer_rsrc_t rsA = mk_rsrc(A + k0 * lda + m0, ((kend - k0) * lda - m0) * 4);
        const __amdgpu_buffer_rsrc_t rsB = mk_rsrc(B + k0 * ldb + n0, ((kend - k0) * ldb - n0) * 4);
        const uint32_t ld4 = (uint32_t)(x.isA ? lda : ldb) * 4u;
        const bool uniformA = __builtin_amdgcn_readfirstlane((int)x.isA) != 0;
        const bool mixed = __builtin_amdgcn_ballot_w64(x.isA) != 0 && __builtin_amdgcn_ballot_w64(!x.isA) != 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint32_t off = (on && x.ok) ? (uint32_t)(x.k8 * 8 + i) * ld4 + (uint32_t)x.c4 * 16u : 0x80000000u;
            if (PROBE & 1) { r[i] = f32x4{1.f, 2.f, 3.f, 4.f}; continue; }
            if (mixed) {
                const f32x4 va = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, (int)(x.isA ? off : 0x80000000u), 0, 0));
                const f32x4 vb = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, (int)(x.isA ? 0x80000000u : off), 0, 0));
                r[i] = x.isA ? va : vb;
            } else if (uniformA) {
                r[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, (int)off, 0, 0));
            } else {
                r[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, (int)off, 0, 0));
            }
        }
    };
    auto sstore1 = [&](const f32x4 (&r)[8], const Item& x) {
        if (PROBE & 2) return;
        unsigned char* img = smem_raw + (x.isA ? 0 : kImgA);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            uint32_t p[4][3];
#pragma unroll
            for (int q = 0; q < 4; ++q) split3_pair(r[2 * q][e], r[2 * q + 1][e], p[q]);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                *reinterpret_cast<uint4*>(img + pl * kPlane + (x.c4 * 4 + e) * ROWB + x.k8 * 16) = make_uint4(p[0][pl], p[1][pl], p[2][pl], p[3][pl]);
        }
    };
    f32x4 acc[MR][NR];
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // BAL: wave w owns reduction rows 8 w .. 8 w + 7 of the stage (a scalar row offset), lane l the columns l + 64 q, q = 0..4, of the 320
    // [A | B] columns: five 8 k x 1 column items per thread, dword loads, one 16-byte piece per plane and item
    float pb[5][8];
    uint32_t voff[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        const int col = lane + 64 * q;
        const bool a = col < BM;
        const int64_t cc = a ? m0 + col : n0 + col - BM;
        voff[q] = cc < (a ? M : N) ? (uint32_t)(a ? col : col - BM) * 4u : 0x80000000u;
    }
    auto gloadb = [&](int kt) {
        const int64_t k0 = kbeg + (int64_t)kt * BKH;
        const __amdgpu_buffer_rsrc_t rsA = mk_rsrc(A + k0 * lda + m0, ((kend - k0) * lda - m0) * 4);
        const __amdgpu_buffer_rsrc_t rsB = mk_rsrc(B + k0 * ldb + n0, ((kend - k0) * ldb - n0) * 4);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int sa = (wid * 8 + i) * (int)lda * 4, sb = (wid * 8 + i) * (int)ldb * 4;
            if (PROBE & 1) {
#pragma unroll
                for (int q = 0; q < 5; ++q) pb[q][i] = 1.f + q;
                continue;
            }
            pb[0][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsA, (int)voff[0], sa, 0));
            pb[1][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsA, (int)voff[1], sa, 0));
            const float xa = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsA, (int)(lane < 32 ? voff[2] : 0x80000000u), sa, 0));
            const float xb = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsB, (int)(lane < 32 ? 0x80000000u : voff[2]), sb, 0));
            pb[2][i] = lane < 32 ? xa : xb;
            pb[3][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsB, (int)voff[3], sb, 0));
            pb[4][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsB, (int)voff[4], sb, 0));
        }
    };
    auto sstoreb = [&]() {
        if (PROBE & 2) return;
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            const int col = lane + 64 * q;
            uint32_t p[4][3];
#pragma unroll
            for (int h = 0; h < 4; ++h) split3_pair(pb[q][2 * h], pb[q][2 * h + 1], p[h]);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)      // (the B image follows the A image: row `col` of the [A | B] column list is row `col` of a plane)
                *reinterpret_cast<uint4*>(smem_raw + pl * kPlane + col * ROWB + wid * 16) = make_uint4(p[0][pl], p[1][pl], p[2][pl], p[3][pl]);
        }
    };
    if constexpr (BAL) {
        gloadb(0);
    } else {
        gload1(p0, it0, 0, true);
        if (two) gload1(p1, it1, 0, true);
    }
#pragma unroll 1
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();
        if constexpr (BAL) {
            sstoreb();
            gloadb(kt + 1);
        } else {
            sstore1(p0, it0);
            if (two) sstore1(p1, it1);
            gload1(p0, it0, kt + 1, true);
            if (two) gload1(p1, it1, kt + 1, true);
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        if (!(PROBE & 4)) {
            const unsigned char* As = smem_raw;
            const unsigned char* Bs = smem_raw + kImgA;
            constexpr int GI = 3;
#pragma unroll
            for (int i0 = 0; i0 < MR; i0 += GI) {
                bf16x8 af[GI][3];
#pragma unroll
                for (int i = 0; i < GI; ++i)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
                        if (i0 + i < MR) af[i][pl] = *reinterpret_cast<const bf16x8*>(As + pl * kPlane + (wm * (BM / 2) + (i0 + i) * 16 + li) * ROWB + lg * 16);
#pragma unroll
                for (int j = 0; j < NR; ++j) {
                    bf16x8 bf[3];
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
                        bf[pl] = *reinterpret_cast<const bf16x8*>(Bs + pl * kPlane + (wn * (BN / 2) + j * 16 + li) * ROWB + lg * 16);
#define X3_TERM(PB, PA)                                                                                         \
    _Pragma("unroll") for (int i = 0; i < GI; ++i) if (i0 + i < MR) acc[i0 + i][j] =                            \
        __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[PB], af[i][PA], acc[i0 + i][j], 0, 0, 0);
                    X3_TERM(0, 2) X3_TERM(2, 0) X3_TERM(1, 1) X3_TERM(0, 1) X3_TERM(1, 0) X3_TERM(0, 0)
#undef X3_TERM
                }
            }
        }
    }
    float* Wz = W + (int64_t)z * M * ldw;
    const int64_t n_store = (N + 3) & ~(int64_t)3;
#pragma unroll
    for (int i = 0; i < MR; ++i) {
        const int64_t row = m0 + wm * (BM / 2) + i * 16 + li;
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const int64_t col0 = n0 + wn * (BN / 2) + j * 16 + lg * 4;
            f32x4 x = acc[i][j];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (col0 + q >= N) x[q] = 0.f;
            if (row < M && col0 < n_store) *reinterpret_cast<f32x4*>(Wz + row * ldw + col0) = x;
        }
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

static int g_loop_reps = 0, g_ordinal = 0, g_only = -1;          // `x3_tn loop <ordinal>`: one variant, 3,000 launches (tools/clock_watch.py)

template <int BM, int BN, int PF, int PROBE, int IG = 0, int V2 = 0>
static void run(const char* name, const float* dH, int M, const float* dG, int N, int64_t K, float* dW, float* dC, const std::vector<float>& hH,
                const std::vector<float>& hG) {
    const int my = g_ordinal++;
    if (g_only >= 0 && my != g_only) return;
    const int n_mt = (M + BM - 1) / BM, n_nt = (N + BN - 1) / BN;
    const int tiles = n_mt * n_nt;
    int ns = 256 / tiles;
    if (ns >= kNumXCD) ns = ns / kNumXCD * kNumXCD;
    const int64_t kchunk = ((K + ns - 1) / ns + BKH - 1) / BKH * BKH;
    const int nsplit = (int)((K + kchunk - 1) / kchunk);
    const int grid = tiles * ((nsplit + kNumXCD - 1) / kNumXCD) * kNumXCD;
    const int64_t ldw = (N + 3) & ~3;
    const int lds = V2 == 1 ? 2 * 3 * (BM + BN) * ROWB2 : 3 * (BM + BN) * ROWB;
    void (*kern)(const float*, int64_t, int64_t, const float*, int64_t, int64_t, int64_t, float*, int64_t, int, int, int, int64_t);
    if constexpr (V2 == 2) kern = x3_tn3_kernel<BM, BN, IG, PROBE, PF>;          // (PF carries ORDER)
    else if constexpr (V2 == 1) kern = x3_tn2_kernel<BM, BN, PF, PROBE>;
    else kern = x3_tn_kernel<BM, BN, PF, PROBE, IG>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, 0, dH, (int64_t)M, (int64_t)M, dG, (int64_t)N, (int64_t)N, K, dW, ldw, n_mt, n_nt, nsplit, kchunk);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int reps = g_loop_reps ? g_loop_reps : 20;
    CK(hipEventRecord(e0));
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, 0, dH, (int64_t)M, (int64_t)M, dG, (int64_t)N, (int64_t)N, K, dW, ldw, n_mt, n_nt, nsplit, kchunk);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    hipLaunchKernelGGL(reduce_kernel, dim3((M * N + 255) / 256), dim3(256), 0, 0, M, N, nsplit, dW, ldw, dC);
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
    printf("[%2d] %-44s grid %4d (%d x %d tiles x %d slabs) lds %6d  %.3f ms  %.1f TF(fp32-equiv)   max err / sum|terms| %.2e\n", my, name, grid, n_mt, n_nt,
           nsplit, lds, ms, 2.0 * M * N * K / ms / 1e9, worst);
}

template <int PROBE, int BAL = 0>
static void run4(const char* name, const float* dH, int M, const float* dG, int N, int64_t K, float* dW, float* dC, const std::vector<float>& hH,
                 const std::vector<float>& hG) {
    const int my = g_ordinal++;
    if (g_only >= 0 && my != g_only) return;
    constexpr int BM = 160, BN = 160;
    const int n_mt = (M + BM - 1) / BM, n_nt = (N + BN - 1) / BN;
    const int tiles = n_mt * n_nt;
    int ns = 512 / tiles;                          // two blocks per CU
    if (ns >= kNumXCD) ns = ns / kNumXCD * kNumXCD;
    const int64_t kchunk = ((K + ns - 1) / ns + BKH - 1) / BKH * BKH;
    const int nsplit = (int)((K + kchunk - 1) / kchunk);
    const int grid = tiles * ((nsplit + kNumXCD - 1) / kNumXCD) * kNumXCD;
    const int64_t ldw = (N + 3) & ~3;
    const int lds = 3 * (BM + BN) * ROWB;
    auto kern = x3_tn4_kernel<PROBE, BAL>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, dH, (int64_t)M, (int64_t)M, dG, (int64_t)N, (int64_t)N, K, dW, ldw, n_mt, n_nt, nsplit, kchunk);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int reps = g_loop_reps ? g_loop_reps : 20;
    CK(hipEventRecord(e0));
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, dH, (int64_t)M, (int64_t)M, dG, (int64_t)N, (int64_t)N, K, dW, ldw, n_mt, n_nt, nsplit, kchunk);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    hipLaunchKernelGGL(reduce_kernel, dim3((M * N + 255) / 256), dim3(256), 0, 0, M, N, nsplit, dW, ldw, dC);
    std::vector<float> hc((size_t)M * N);
    CK(hipMemcpy(hc.data(), dC, hc.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0;
    const int ms_[] = {0, 1, 15, 16, 79, 80, 159, 160, 299}, ns_[] = {0, 17, 159, 160, 299, 300, 319, 320, 599};
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
    printf("[%2d] %-44s grid %4d (%d x %d tiles x %d slabs) lds %6d  %.3f ms  %.1f TF(fp32-equiv)   max err / sum|terms| %.2e\n", my, name, grid, n_mt, n_nt,
           nsplit, lds, ms, 2.0 * M * N * K / ms / 1e9, worst);
}

int main(int argc, char** argv) {
    if (argc >= 3 && !strcmp(argv[1], "loop")) { g_only = atoi(argv[2]); g_loop_reps = 3000; }
    const int64_t K = 440000;
    const int M = 300;
    for (int N : {600, 300, 256}) {
        std::vector<float> hH((size_t)K * M), hG((size_t)K * N);
        uint32_t s = 12345;
        auto rnd = [&]() {
            s = s * 1664525u + 1013904223u;
            return (float)((s >> 8) & 0xffff) / 65536.f - 0.5f;
        };
        for (auto& x : hH) x = rnd() * (1.f + 1e-3f * rnd());
        for (auto& x : hG) x = rnd() * 0.1f * (1.f + 1e-3f * rnd());
        float *dH, *dG, *dW, *dC;
        CK(hipMalloc(&dH, hH.size() * 4));
        CK(hipMalloc(&dG, hG.size() * 4));
        CK(hipMalloc(&dW, (size_t)256 * M * 640 * 4));
        CK(hipMalloc(&dC, (size_t)M * N * 4));
        CK(hipMemcpy(dH, hH.data(), hH.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dG, hG.data(), hG.size() * 4, hipMemcpyHostToDevice));
        printf("M = %d, N = %d, K = %lld (%.1f GFLOP)\n", M, N, (long long)K, 2.0 * M * N * K / 1e9);
        run<160, 320, 1, 0>("x3 A^T.B, 160 x 320, 1 stage ahead", dH, M, dG, N, K, dW, dC, hH, hG);
        run<160, 320, 2, 0>("x3 A^T.B, 160 x 320, 2 stages ahead", dH, M, dG, N, K, dW, dC, hH, hG);
        run<160, 320, 0, 0, 3, 2>("v3: split in registers, halves ping-pong, GI 3", dH, M, dG, N, K, dW, dC, hH, hG);
        run<160, 320, 0, 0, 0, 2>("v3: split in registers, halves ping-pong, GI 5", dH, M, dG, N, K, dW, dC, hH, hG);
        run<160, 320, 1, 0, 3, 2>("v3: every wave splits first, GI 3", dH, M, dG, N, K, dW, dC, hH, hG);
        run<160, 320, 2, 0, 3, 2>("v3: every wave multiplies first, GI 3", dH, M, dG, N, K, dW, dC, hH, hG);
        run<160, 320, 0, 1, 3, 2>("  v3 ablation: no global loads", dH, M, dG, N, K, dW, dC, hH, hG);
        run<160, 320, 0, 3, 3, 2>("  v3 ablation: MFMAs + fragment reads only", dH, M, dG, N, K, dW, dC, hH, hG);
        run<160, 320, 0, 4, 3, 2>("  v3 ablation: no MFMAs", dH, M, dG, N, K, dW, dC, hH, hG);
        if (N <= 256) run<160, 256, 0, 0, 3, 2>("v3: 160 x 256, ping-pong, GI 3", dH, M, dG, N, K, dW, dC, hH, hG);
        run4<0>("v4: two blocks per CU, 160 x 160", dH, M, dG, N, K, dW, dC, hH, hG);
        run4<1>("  v4 ablation: no global loads", dH, M, dG, N, K, dW, dC, hH, hG);
        run4<3>("  v4 ablation: MFMAs + fragment reads only", dH, M, dG, N, K, dW, dC, hH, hG);
        run4<4>("  v4 ablation: no MFMAs", dH, M, dG, N, K, dW, dC, hH, hG);
        run4<0, 1>("v5: two blocks per CU, balanced column items", dH, M, dG, N, K, dW, dC, hH, hG);
        run4<1, 1>("  v5 ablation: no global loads", dH, M, dG, N, K, dW, dC, hH, hG);
        run4<3, 1>("  v5 ablation: MFMAs + fragment reads only", dH, M, dG, N, K, dW, dC, hH, hG);
        run4<4, 1>("  v5 ablation: no MFMAs", dH, M, dG, N, K, dW, dC, hH, hG);
        run<160, 320, 2, 1>("  ablation: no global loads", dH, M, dG, N, K, dW, dC, hH, hG);
        run<160, 320, 2, 3>("  ablation: MFMAs + fragment reads only", dH, M, dG, N, K, dW, dC, hH, hG);
        run<160, 320, 2, 4>("  ablation: no MFMAs", dH, M, dG, N, K, dW, dC, hH, hG);
        if (N <= 256) run<160, 256, 2, 0>("x3 A^T.B, 160 x 256, 2 stages ahead", dH, M, dG, N, K, dW, dC, hH, hG);
        run<128, 320, 2, 0>("x3 A^T.B, 128 x 320, 2 stages ahead", dH, M, dG, N, K, dW, dC, hH, hG);
        CK(hipFree(dH)); CK(hipFree(dG)); CK(hipFree(dW)); CK(hipFree(dC));
    }
    return 0;
}
