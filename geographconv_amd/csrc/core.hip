// ABI version + thread-local error text for libgeogcn.so.
#include "common.h"

#include <stdlib.h>

#include <algorithm>

#include <stdarg.h>

namespace geogcn {
namespace {
thread_local char g_err[512] = "";
}

int64_t test_seam_i64(const char* name, int64_t dflt) {
    const char* e = getenv(name);
    if (!e || !*e) return dflt;
    char* end = nullptr;
    const long long v = strtoll(e, &end, 10);
    return (end && *end == 0 && v > 0) ? (int64_t)v : dflt;
}

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

namespace {
__global__ __launch_bounds__(256) void zero_fill_kernel(uint32_t* __restrict__ p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = 0u;
}

__global__ __launch_bounds__(256) void zero_rows_kernel(float* __restrict__ p, int64_t rows, int64_t cols, int64_t ld) {
    const int64_t total = rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256)
        p[(i / cols) * ld + i % cols] = 0.f;
}
}  // namespace

int zero_rows_async(float* ptr, int64_t rows, int64_t cols, int64_t ld, hipStream_t st) {
    if (rows <= 0 || cols <= 0) return 0;
    const unsigned grid = (unsigned)std::min<int64_t>((rows * cols + 255) / 256, (int64_t)kNumCU * 16);
    hipLaunchKernelGGL(zero_rows_kernel, dim3(grid), dim3(256), 0, st, ptr, rows, cols, ld);
    GEOGCN_LAUNCH_CHECK("zero_rows_kernel");
    return 0;
}

int zero_fill_async(void* ptr, size_t bytes, hipStream_t st) {
    if (bytes == 0) return 0;
    const size_t n = bytes / 4;
    const unsigned grid = (unsigned)std::min<size_t>((n + 255) / 256, (size_t)kNumCU * 16);
    hipLaunchKernelGGL(zero_fill_kernel, dim3(grid), dim3(256), 0, st, (uint32_t*)ptr, n);
    GEOGCN_LAUNCH_CHECK("zero_fill_kernel");
    return 0;
}
}  // namespace geogcn

extern "C" {
int geogcn_version(void) { return GEOGCN_ABI_VERSION; }
const char* geogcn_last_error(void) { return geogcn::g_err; }
}
