// Probe (NOT part of the product): which physical CUs does a hipExtStreamCreateWithCUMask bit pattern select on a
// gfx950 (8 XCDs x 32 CUs)?  Every workgroup records (XCC id, SE id, CU id) of the CU it runs on; the host prints, per
// mask pattern, how many distinct CUs were seen in each XCD.   hipcc --offload-arch=gfx950 -O2 cu_mask_map.hip -o bin/cu_mask_map
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <set>
#include <vector>

__global__ void where_kernel(uint32_t* out, int spin) {
    uint32_t xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    // keep the CU busy for a while so that the blocks of one launch spread over every CU the mask allows
    uint64_t t0 = wall_clock64();
    while (wall_clock64() - t0 < (uint64_t)spin) {}
    if (threadIdx.x == 0) out[blockIdx.x] = ((xcc & 0xf) << 16) | (hw & 0xffff);
}

static void run(const char* name, const std::vector<int>& bits) {
    uint32_t words[8] = {0};
    for (int b : bits) words[b / 32] |= 1u << (b % 32);
    hipStream_t st;
    if (hipExtStreamCreateWithCUMask(&st, 8, words) != hipSuccess) { printf("%s: stream creation failed\n", name); return; }
    const int nb = 4096;
    uint32_t* d;
    hipMalloc(&d, nb * 4);
    hipMemset(d, 0xff, nb * 4);
    where_kernel<<<nb, 64, 0, st>>>(d, 20000);
    hipStreamSynchronize(st);
    std::vector<uint32_t> h(nb);
    hipMemcpy(h.data(), d, nb * 4, hipMemcpyDeviceToHost);
    std::set<uint32_t> per_xcc[16];
    for (uint32_t v : h) per_xcc[(v >> 16) & 0xf].insert(v & 0xff00);     // HW_ID: cu_id [11:8], sh_id [12], se_id [15:13]
    printf("%-34s bits=%3zu  CUs seen per XCD:", name, bits.size());
    int total = 0;
    for (int x = 0; x < 8; ++x) { printf(" %2zu", per_xcc[x].size()); total += (int)per_xcc[x].size(); }
    printf("   total %d\n", total);
    hipFree(d);
    hipStreamDestroy(st);
}

int main() {
    auto range = [](int a, int b, int step = 1) { std::vector<int> v; for (int i = a; i < b; i += step) v.push_back(i); return v; };
    run("all 256", range(0, 256));
    run("first 192", range(0, 192));
    run("first 128", range(0, 128));
    run("first 64", range(0, 64));
    run("first 32", range(0, 32));
    run("first 8", range(0, 8));
    run("bits 8..15", range(8, 16));
    run("last 64 (192..255)", range(192, 256));
    run("every 2nd", range(0, 256, 2));
    run("every 4th", range(0, 256, 4));
    run("every 8th", range(0, 256, 8));
    run("every 8th from 1", range(1, 256, 8));
    run("bits 0..31 step 2", range(0, 32, 2));
    return 0;
}
