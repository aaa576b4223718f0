/* CPU ORACLE, multi-threaded leg -- TEST / BENCH INFRASTRUCTURE ONLY (never linked into libgeogcn.so, never imported by
 * geographconv_amd/).  The "fair multi-core CPU" baseline SURVEY.md section 8d asks for next to the Theano-equivalent one:
 * the same CSR x dense product as oracle/gcn_oracle.py::spmm (reference gcnmodel.py:39,130,153: S.structured_dot; Theano
 * runs it as ONE single-threaded C loop over the rows), here with the rows spread over all host cores by OpenMP.
 * Accumulation per row is sequential in stored index order in fp32 -- the single-threaded loop's order; the compiler
 * may contract a*b + c into fma, so results equal scipy's csr @ dense to an ulp or two (tests/test_oracle.py).
 *   gcc -O3 -march=native -fopenmp -shared -fPIC oracle/cpu_mt.c -o oracle/_build/libcpu_mt.so                       */
#include <omp.h>
#include <stdint.h>
#include <string.h>

int geogcn_cpu_threads(void) { return omp_get_max_threads(); }

/* C[n_rows x F] = A_csr . B   (row-major, ldb / ldc in elements) */
void geogcn_cpu_spmm_f32(int64_t n_rows, const int32_t* rowptr, const int32_t* colidx, const float* val, const float* B,
                         int64_t ldb, float* C, int64_t ldc, int64_t F) {
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t r = 0; r < n_rows; ++r) {
        float* c = C + r * ldc;
        memset(c, 0, (size_t)F * sizeof(float));
        for (int32_t j = rowptr[r]; j < rowptr[r + 1]; ++j) {
            const float a = val[j];
            const float* b = B + (int64_t)colidx[j] * ldb;
            for (int64_t k = 0; k < F; ++k) c[k] += a * b[k];
        }
    }
}
