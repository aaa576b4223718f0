/* CPU ORACLE, multi-threaded leg -- TEST / BENCH INFRASTRUCTURE ONLY (never linked into libgeogcn.so, never imported by
 * geographconv_amd/).  The "fair multi-core CPU" baseline SURVEY.md section 8d asks for next to the Theano-equivalent one:
 * every pass of one f_train step that is not a BLAS call, spread over all host cores by OpenMP --
 *   - the CSR x dense product of oracle/gcn_oracle.py::spmm (reference gcnmodel.py:39,130,153: S.structured_dot; Theano runs
 *     it as ONE single-threaded C loop over the rows): rows over threads, accumulation per row sequential in stored index
 *     order in fp32, i.e. the single-threaded loop's order;
 *   - the fused elementwise passes Theano's Elemwise fusion produces (bias + tanh, the highway mix of gcnmodel.py:266 with the
 *     gate's sigmoid, their gradients), row softmax (gcnmodel.py:374), the scattered cross-entropy gradient, column sums
 *     (bias gradients: per-thread partial sums over contiguous row blocks, added in thread order).
 * The dense products stay on BLAS sgemm (NumPy), which is multi-threaded in both legs.  oracle/cpu_mt.py::f_train strings these
 * together; tests/test_oracle.py holds it to oracle.f_train.  The compiler may contract a*b + c into fma, so results equal
 * the NumPy restatement to rounding, not bit for bit.
 *   gcc -O3 -march=native -fopenmp -shared -fPIC oracle/cpu_mt.c -o oracle/_build/libcpu_mt.so -lm                     */
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
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

static inline float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

/* Y = tanh(S + b), and (scale given) Yd = Y * scale          gcnmodel.py:41-42,132-136; lasagne DropoutLayer :357 */
void geogcn_cpu_bias_tanh_f32(int64_t n, int64_t F, const float* S, const float* b, float* Y, const float* scale, float* Yd) {
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < n; ++r)
        for (int64_t k = 0; k < F; ++k) {
            const float y = tanhf(S[r * F + k] + b[k]);
            Y[r * F + k] = y;
            if (scale) Yd[r * F + k] = y * scale[r * F + k];
        }
}

/* Hc = tanh(S + bh); T = sigmoid(U + bt); Hout = T * Hc + (1 - T) * H          gcnmodel.py:136,286,266 */
void geogcn_cpu_highway_fwd_f32(int64_t n, int64_t F, const float* S, const float* bh, const float* U, const float* bt,
                                const float* H, float* Hc, float* T, float* Hout) {
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < n; ++r)
        for (int64_t k = 0; k < F; ++k) {
            const int64_t i = r * F + k;
            const float hc = tanhf(S[i] + bh[k]), t = sigmoidf_(U[i] + bt[k]);
            Hc[i] = hc;
            T[i] = t;
            Hout[i] = t * hc + (1.0f - t) * H[i];
        }
}

/* P = softmax_rows(S + b)          gcnmodel.py:155-157 */
void geogcn_cpu_bias_softmax_f32(int64_t n, int64_t C, const float* S, const float* b, float* logits, float* P) {
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < n; ++r) {
        float m = -INFINITY;
        for (int64_t k = 0; k < C; ++k) {
            const float x = S[r * C + k] + b[k];
            logits[r * C + k] = x;
            m = x > m ? x : m;
        }
        float s = 0.f;
        for (int64_t k = 0; k < C; ++k) {
            const float e = expf(logits[r * C + k] - m);
            P[r * C + k] = e;
            s += e;
        }
        for (int64_t k = 0; k < C; ++k) P[r * C + k] /= s;
    }
}

/* D = 0; D[idx[j], :] += (P[idx[j], :] - onehot(y[j])) * inv_n          gradient of gcnmodel.py:376,382 (AdvancedIncSubtensor1) */
void geogcn_cpu_ce_grad_f32(int64_t n, int64_t C, const float* P, const int64_t* idx, const int64_t* y, int64_t n_idx, float inv_n,
                            float* D) {
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < n; ++r) memset(D + r * C, 0, (size_t)C * sizeof(float));
    /* (rows may repeat: sequential over j then, as np.add.at; the reference's index vectors are unique, gcnmain.py:207) */
    for (int64_t j = 0; j < n_idx; ++j) {
        const int64_t r = idx[j];
        for (int64_t k = 0; k < C; ++k) D[r * C + k] += (P[r * C + k] - (k == y[j] ? 1.0f : 0.0f)) * inv_n;
    }
}

/* per-thread partial column sums over contiguous row blocks, combined in thread order */
static void colsum2_(int64_t n, int64_t F, const float* X0, const float* X1, float* s0, float* s1) {
    const int nt = omp_get_max_threads();
    float* part = (float*)calloc((size_t)nt * 2 * F, sizeof(float));
    /* the nt row blocks are loop iterations, not thread ids: every block is summed exactly once whatever team the runtime
     * delivers (OMP_DYNAMIC, OMP_THREAD_LIMIT, cgroup limits, nesting may give fewer threads than omp_get_max_threads) */
#pragma omp parallel for schedule(static)
    for (int t = 0; t < nt; ++t) {
        const int64_t lo = n * t / nt, hi = n * (t + 1) / nt;
        float* p0 = part + (size_t)t * 2 * F;
        float* p1 = p0 + F;
        for (int64_t r = lo; r < hi; ++r)
            for (int64_t k = 0; k < F; ++k) {
                p0[k] += X0[r * F + k];
                if (X1) p1[k] += X1[r * F + k];
            }
    }
    for (int64_t k = 0; k < F; ++k) {
        float a = 0.f, b = 0.f;
        for (int t = 0; t < nt; ++t) {
            a += part[(size_t)t * 2 * F + k];
            b += part[(size_t)t * 2 * F + F + k];
        }
        s0[k] = a;
        if (s1) s1[k] = b;
    }
    free(part);
}

void geogcn_cpu_colsum_f32(int64_t n, int64_t F, const float* X, float* s) { colsum2_(n, F, X, NULL, s, NULL); }

/* dS = G * T * (1 - Hc^2); dU = G * (Hc - H) * T * (1 - T); dH = G * (1 - T); dbh = colsum(dS); dbt = colsum(dU)
 * (what autodiff derives for gcnmodel.py:266 through tanh :136 and sigmoid :286) */
void geogcn_cpu_highway_bwd_f32(int64_t n, int64_t F, const float* G, const float* T, const float* Hc, const float* H, float* dS,
                                float* dU, float* dH, float* dbh, float* dbt) {
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < n; ++r)
        for (int64_t k = 0; k < F; ++k) {
            const int64_t i = r * F + k;
            const float g = G[i], t = T[i], hc = Hc[i];
            dS[i] = (g * t) * (1.0f - hc * hc);
            dU[i] = ((g * (hc - H[i])) * t) * (1.0f - t);
            dH[i] = g * (1.0f - t);
        }
    colsum2_(n, F, dS, dU, dbh, dbt);
}

/* dS0 = (G [* scale]) * (1 - H0^2); db0 = colsum(dS0)          gradient of gcnmodel.py:42 (and of the dropout of :357) */
void geogcn_cpu_tanh_bwd_f32(int64_t n, int64_t F, const float* G, const float* scale, const float* H0, float* dS0, float* db0) {
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < n; ++r)
        for (int64_t k = 0; k < F; ++k) {
            const int64_t i = r * F + k;
            const float g = scale ? G[i] * scale[i] : G[i];
            dS0[i] = g * (1.0f - H0[i] * H0[i]);
        }
    colsum2_(n, F, dS0, NULL, db0, NULL);
}

/* C = A + B (the two halves of dH) */
void geogcn_cpu_add_f32(int64_t n, const float* A, const float* B, float* C) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) C[i] = A[i] + B[i];
}
