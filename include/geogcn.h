/* geogcn.h -- C ABI of libgeogcn.so: the MI355X (gfx950) replacement for the native code the
 * reference's GCN hot path runs in (the C that Theano 1.0.x generates for the ops composed by
 * /root/reference/gcnmodel.py).  One entry point per Theano op family the path exercises; each
 * comment cites the reference call site the entry point replaces.
 *
 * Conventions
 *   - every function returns 0 on success, <0 for an argument error (GEOGCN_E_*), >0 for a
 *     hipError_t passed through;  geogcn_last_error() gives the text (thread-local).
 *   - all data pointers are BORROWED DEVICE pointers (the caller -- a torch tensor on the Python
 *     side -- owns the memory and keeps it alive until the stream is synchronised), except where a
 *     parameter is named *_host.
 *   - `stream` is a hipStream_t passed as void*; every launch goes on it; no hidden syncs.
 *   - dense matrices are row-major fp32 with an explicit leading dimension (`ld*`, in elements).
 *     Vectorised kernels need ld % 4 == 0 and 16-byte aligned bases; other pitches fall back to a
 *     scalar kernel (SpMM) or are rejected with GEOGCN_E_ALIGN (GEMM).  Producers keep the pad
 *     columns [F, ld) ZERO so a padded matrix can be used as a reduction operand.
 *   - CSR: int32 rowptr[n_rows+1], int32 colidx[nnz], fp32 val[nnz] (scipy layout; reference
 *     gcnmain.py:172-179 casts X and A to float32 CSR).
 *   - no C++ exceptions cross this boundary.
 */
#ifndef GEOGCN_H
#define GEOGCN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: round 3 changed exported signatures incompatibly (geogcn_spmm_plan_create: chunks_with_owner instead of a rowsplit pointer;
 * geogcn_gemm_kcat_f32: ws / ws_bytes before stream; geogcn_spmm_plan_attach_timer replaced geogcn_timer_attach_spmm);
 * round 4: geogcn_spmm_csr_hot_f32 / _hot_dropout_f32 take n_cols after n_rows and row_order after n_hot.
 * Added since without a version change (nothing existing changed meaning): geogcn_gemm_kcat_gated_f32, geogcn_gemm_gated_f32,
 * geogcn_gemm_kcat_gated_tanhbwd_f32, geogcn_gemm_dual_bf16 (+ _workspace_bytes), geogcn_gate_carry_f32, geogcn_colsum_rowblocks_f32,
 * geogcn_spmm_csr_softmax_f32;
 * geogcn_highway_bwd_f32 / _bf16s_f32 accept dHcarry = NULL.
 * 3: round 5: the fused highway launches take `precision` (GEOGCN_GEMM_F32 or GEOGCN_GEMM_BF16X3) before their workspace --
 * geogcn_gemm_dual_f32, geogcn_gemm_kcat_f32, geogcn_gemm_kcat_gated_f32, geogcn_gemm_kcat_gated_tanhbwd_f32 and the two
 * _workspace_bytes functions that size it (as their last argument); GEOGCN_GEMM_BF16X3 now also applies to transA = 1. */
#define GEOGCN_ABI_VERSION 3

#define GEOGCN_E_NULL   (-1)   /* required pointer is NULL            */
#define GEOGCN_E_SIZE   (-2)   /* negative / inconsistent size        */
#define GEOGCN_E_ALIGN  (-3)   /* pitch or base alignment unsupported */
#define GEOGCN_E_ARG    (-4)   /* bad enum / flag / workspace too small */

/* epilogue selectors (lasagne.nonlinearities used on the path: gcnmodel.py:286,347,374) */
#define GEOGCN_ACT_NONE    0   /* linear (softmax layers keep logits; softmax is its own entry) */
#define GEOGCN_ACT_TANH    1   /* lasagne.nonlinearities.tanh     gcnmodel.py:347 */
#define GEOGCN_ACT_SIGMOID 2   /* T.nnet.sigmoid                  gcnmodel.py:286 */
/* elementwise entry points only (geogcn_bias_act_f32 / geogcn_act_bwd_f32), not fused epilogues: */
#define GEOGCN_ACT_SELU    3   /* lasagne.nonlinearities.selu     gcnmodel.py:290 (residual_dense) */
#define GEOGCN_ACT_RELU    4   /* lasagne.nonlinearities.rectify  gcnmodel.py:345 (commented out)  */

int         geogcn_version(void);
const char* geogcn_last_error(void);

/* ---- K1/K2/K3/K4: S.structured_dot(CSR, dense) ---------------------------------------------
 * C[n_rows x F] = act(A_csr . B + bias)            gcnmodel.py:39 (X.W0), :130, :153 (A_hat.Z)
 * and, called on CSR(A^T), the gradient Theano derives for them (StructuredDot grad).
 * Rows are accumulated in stored index order; rows longer than the plan's threshold are split
 * into fixed chunks whose partial sums are combined in chunk order (deterministic, no atomics).
 * bias may be NULL.  plan may be NULL (no row splitting: correct, slow on hub rows).          */
typedef struct geogcn_spmm_plan geogcn_spmm_plan;

/* chunks_with_owner: where the chunks of the long rows run (speed only, never a value).  1 = on the XCD that owns the
 * long row, at the row's place in that XCD's sweep -- for node numberings with locality, where a hub's neighbours share
 * its L2 window; 0 = dealt round the XCDs, ahead of the row blocks -- for numberings without (e.g. hubs first).      */
int    geogcn_spmm_plan_create(int32_t n_rows, const int32_t* rowptr_host, int32_t long_row_nnz, int32_t chunk_nnz,
                               int32_t chunks_with_owner, geogcn_spmm_plan** out);
void   geogcn_spmm_plan_destroy(geogcn_spmm_plan* plan);
int64_t geogcn_spmm_plan_num_long_rows(const geogcn_spmm_plan* plan);
int64_t geogcn_spmm_plan_num_chunks(const geogcn_spmm_plan* plan);
size_t geogcn_spmm_workspace_bytes(const geogcn_spmm_plan* plan, int32_t F);

int geogcn_spmm_csr_f32(const geogcn_spmm_plan* plan, int32_t n_rows, int32_t n_cols, int64_t nnz,
                        const int32_t* rowptr, const int32_t* colidx, const float* val,
                        const float* B, int64_t ldb, float* C, int64_t ldc, int32_t F,
                        const float* bias, int32_t act, void* ws, size_t ws_bytes, void* stream);

/* P = softmax(A . B + bias) row by row, argmax_out (nullable) = the first index of each row's maximum: the output layer's graph
 * product (gcnmodel.py:149) and its softmax nonlinearity in one launch -- the logits are never written.  32 < F <= 512; other widths:
 * geogcn_spmm_csr_f32 + geogcn_softmax_rows_f32.  Same maximum, index and exponentials as that pair; the row sum is taken over this
 * kernel's layout (the probabilities agree to rounding, not bit for bit).                                                    */
int geogcn_spmm_csr_softmax_f32(const geogcn_spmm_plan* plan, int32_t n_rows, int32_t n_cols, int64_t nnz,
                                const int32_t* rowptr, const int32_t* colidx, const float* val, const float* B, int64_t ldb,
                                float* P, int64_t ldp, int32_t F, const float* bias, int32_t* argmax_out, void* ws,
                                size_t ws_bytes, void* stream);
/* C = act(C + A_csr . B + bias): the accumulators start from the row already in C (a product that continues one begun
 * by another kernel -- X.W0 as dense head panel on the MFMA pipe + CSR tail, gcnmodel.py:39).  Same kernel, same order
 * (C's value first, then the stored nonzeros in index order).  Needs float4-addressable operands (GEOGCN_E_ALIGN).  */
int geogcn_spmm_csr_acc_f32(const geogcn_spmm_plan* plan, int32_t n_rows, int32_t n_cols, int64_t nnz,
                            const int32_t* rowptr, const int32_t* colidx, const float* val, const float* B, int64_t ldb,
                            float* C, int64_t ldc, int32_t F, const float* bias, int32_t act, void* ws, size_t ws_bytes,
                            void* stream);

/* C = act(A_csr . B + bias) with the HOT rows of B staged in LDS -- S.structured_dot(X, W0), gcnmodel.py:39-42, for a
 * bag-of-words X whose columns are Zipfian (the ~120 most frequent vocabulary entries hold more than half of the stored
 * nonzeros at the TwitterUS shape): one persistent workgroup per CU copies the n_hot rows B[hot_rows[s]] into its LDS
 * once and serves the hot nonzeros from there; only the cold ones are gathered from L2 / HBM.  The caller prepares the
 * structure: every row is ordered [hot | cold] (each part in ascending column order), rowsplit[r] is the boundary, and
 * colidx of a HOT entry holds its LDS slot s (0 <= s < n_hot) instead of the column.  n_hot <= geogcn_spmm_hot_capacity(F)
 * (0 = this width is not supported: use geogcn_spmm_csr_f32).  Accumulation is sequential in the stored order.
 * n_cols = rows of B (every cold column index is < n_cols): when n_cols * ldb * 4 < 2^31 the cold rows are gathered through
 * one buffer descriptor with 32-bit offsets, four rows in flight per 16-lane group; 0 = unknown (64-bit addresses, two).
 * row_order (nullable; a permutation of [0, n_rows)): the ORDER in which rows are worked on -- speed only, never a value: the four
 * 16-lane groups of a wave run in lockstep, so a wave takes as long as the longest of its four rows; the caller can put rows of
 * similar cold length next to each other (workgroup b of G works on positions [n_rows b / G, n_rows (b+1) / G), four per wave). */
int32_t geogcn_spmm_hot_capacity(int32_t F);
int geogcn_spmm_csr_hot_f32(int32_t n_rows, int32_t n_cols, const int32_t* rowptr, const int32_t* rowsplit, const int32_t* colidx,
                            const float* val, const float* B, int64_t ldb, const int32_t* hot_rows, int32_t n_hot,
                            const int32_t* row_order, float* C, int64_t ldc, int32_t F, const float* bias, int32_t act, void* stream);

/* The same product with the dropout that FOLLOWS this layer in the reference (lasagne.layers.dropout, gcnmodel.py:357)
 * in its epilogue:  C = act(A . B + bias)  (the layer output, which the backward differentiates through),
 * Cd = C * keep / (1 - p_drop)  (what the next layer reads; same pitch ldc) and the keep-mask bytes.  The keep decisions
 * are those of geogcn_dropout_mask_philox(n_rows, F, p_drop, seed, offset) -- bit-identical -- or, with calls_dev given,
 * of geogcn_dropout_mask_philox_ctr; with mask_in given they are READ from it instead (parity runs inject the mask).
 * mask_out (nullable when mask_in is given) receives the mask used, dense [n_rows][F].  One launch replaces the mask
 * kernel and an apply pass that re-reads C.  Needs F % 4 == 0 (GEOGCN_E_ALIGN otherwise: use the separate calls);
 * act is tanh or none.                                                                                           */
int geogcn_spmm_csr_hot_dropout_f32(int32_t n_rows, int32_t n_cols, const int32_t* rowptr, const int32_t* rowsplit, const int32_t* colidx,
                                    const float* val, const float* B, int64_t ldb, const int32_t* hot_rows, int32_t n_hot,
                                    const int32_t* row_order, float* C, float* Cd, int64_t ldc, int32_t F, const float* bias, int32_t act,
                                    float p_drop,
                                    const uint8_t* mask_in, uint8_t* mask_out, uint64_t seed, uint64_t offset,
                                    const int64_t* calls_dev, int64_t per_call_elems, int64_t base_elems, void* stream);

/* dW[n_words x F] = X^T . G  -- the gradient of S.structured_dot(X, W0) w.r.t. W0 (autodiff of gcnmodel.py:39) for a
 * bag-of-words X whose transpose is given as CSR (rows = vocabulary, columns = documents, STRICTLY ASCENDING inside every
 * row: checked by geogcn_xt_plan_create, GEOGCN_E_ARG otherwise -- the sweep and its entry points rely on it).  Instead of gathering
 * rows of the N x F matrix G at random (every row fetched ~nnz/N times from beyond the L2), the documents are
 * partitioned over 8 groups of workgroups (one per XCD) that sweep their range in L2-sized blocks while each 16-lane
 * group accumulates a fixed set of vocabulary rows in LDS; the 8 partial results per row are added in fixed order.
 * Deterministic, no atomics.  The plan (host-built once per X^T and F: row-to-workgroup assignment balanced by
 * nonzeros, per-range entry points) is independent of the VALUES, so value dropout on X re-uses it.
 * Every row of dW is written (rows without nonzeros as zeros; pad columns [F, roundup4(F)) as zeros).           */
typedef struct geogcn_xt_plan geogcn_xt_plan;
int    geogcn_xt_plan_create(int32_t n_words, int32_t n_docs, const int32_t* rowptr_t_host, const int32_t* docidx_t_host,
                             int32_t F, geogcn_xt_plan** out);
void   geogcn_xt_plan_destroy(geogcn_xt_plan* plan);
size_t geogcn_xt_workspace_bytes(const geogcn_xt_plan* plan);
int    geogcn_xt_dot_f32(const geogcn_xt_plan* plan, const int32_t* docidx_t, const float* val_t, const float* G, int64_t ldg,
                         float* dW, int64_t ldw, void* ws, size_t ws_bytes, void* stream);

/* Reduced-precision variant for the bf16 configuration (BASELINE config 5, `-gemm-precision bf16`; no
 * reference counterpart -- the reference is fp32 throughout): the GATHERED operand B is stored as
 * bfloat16 (raw uint16 bit patterns, row pitch ldb ELEMENTS), everything else -- CSR values, products,
 * the sequential fp32 accumulation, bias, activation, C -- is as in geogcn_spmm_csr_f32.  The kernel is
 * bound by the gather traffic from beyond the L2, so halving the row nearly halves its run time.  Requires
 * 16-byte aligned B and C, ldb % 8 == 0 and >= roundup8(F) (pad columns zero), ldc % 4 == 0, F <= 1024
 * (GEOGCN_E_ALIGN otherwise; there is no scalar fallback).  geogcn_cast_bf16_f32 produces B: round-to-nearest-even, whole pitch written
 * (pad columns [F, ldy) as zeros).                                                               */
int geogcn_spmm_csr_bf16b(const geogcn_spmm_plan* plan, int32_t n_rows, int32_t n_cols, int64_t nnz,
                          const int32_t* rowptr, const int32_t* colidx, const float* val,
                          const uint16_t* B, int64_t ldb, float* C, int64_t ldc, int32_t F,
                          const float* bias, int32_t act, void* ws, size_t ws_bytes, void* stream);
int geogcn_cast_bf16_f32(int64_t n, int32_t F, const float* X, int64_t ldx, uint16_t* Y, int64_t ldy, void* stream);

/* Highway block in one launch (gcnmodel.py:130-136 + :266): Hc = tanh(A.B + bias) and
 * Hout = T*Hc + (1-T)*H, with the gate T and the carried input H read row by row in the SpMM's epilogue
 * instead of a separate elementwise pass over Hc.  T, H, Hc, Hout share the pitch `ld` (% 4 == 0, 16-byte
 * bases; pad columns stay zero).  B is fp32 (b_bf16 = 0) or bf16 bit patterns (b_bf16 = 1, ldb in
 * elements, see geogcn_spmm_csr_bf16b).  Element for element the same arithmetic as
 * geogcn_spmm_csr_f32(act = tanh) followed by geogcn_highway_fwd_f32.                                */
int geogcn_spmm_csr_highway_f32(const geogcn_spmm_plan* plan, int32_t n_rows, int32_t n_cols, int64_t nnz,
                                const int32_t* rowptr, const int32_t* colidx, const float* val, const void* B,
                                int64_t ldb, int32_t b_bf16, int32_t F, const float* bias, const float* T,
                                const float* H, int64_t ld, float* Hc, float* Hout, void* ws, size_t ws_bytes,
                                void* stream);

/* profiling aid (bench.py's roofline leg): a pool of hipEvent pairs owned by the CALLER and attached to ONE plan handle
 * (no library-global state: two models, or two threads with their own plans, do not see each other's timer).  While a
 * timer is attached to `plan`, every geogcn_spmm_csr_f32 / _bf16b call on that plan (not the fused _highway one) whose F
 * equals `only_F` (0 = any) records one (begin, end) pair around the whole product (spmm_rows_kernel + the long rows'
 * combine kernel) on the call's stream, until the pool is full.  geogcn_timer_read_ms synchronises the recorded events
 * and returns the per-launch durations.  Detach with geogcn_spmm_plan_attach_timer(plan, NULL, 0) -- before destroying
 * the timer.                                                                                                     */
typedef struct geogcn_timer geogcn_timer;
int  geogcn_timer_create(int32_t capacity, geogcn_timer** out);
void geogcn_timer_destroy(geogcn_timer* t);
int  geogcn_spmm_plan_attach_timer(geogcn_spmm_plan* plan, geogcn_timer* t, int32_t only_F);
int  geogcn_timer_read_ms(geogcn_timer* t, float* out_ms, int32_t max_out, int32_t* n_out);

/* ---- K5/K6: T.dot / Gemm -----------------------------------------------------------------
 * C[M x N] = act(op(A) . op(B) + bias)   fp32 MFMA (v_mfma_f32_16x16x4_f32), fp32 accumulate.
 *   transA=0: A is M x K row-major (lda);  transA=1: A is K x M row-major (C = A^T . B)
 *   transB=0: B is K x N row-major (ldb);  transB=1: B is N x K row-major (C = A . B^T)
 * gcnmodel.py:126,149 (T.dot(input, W)), lasagne DenseLayer gate :285, and the Gemm ops autodiff
 * derives: dW = H^T.dZ (transA=1), dH = dZ.W^T (transB=1).
 * accumulate=1 adds into C (C += ...), used for dH += dZ.Wh^T + dU.Wt^T.
 * Operands are read in whole float4s: when a contiguous dimension (K for a non-transposed A or a
 * transposed B) is not a multiple of 4, its pad columns up to the next multiple of 4 MUST be zero
 * (the convention every producer in this library keeps).  C is written in whole float4s as well:
 * A, B and C need 16-byte aligned bases and leading dimensions that are multiples of 4 (GEOGCN_E_ALIGN
 * otherwise); C's pad columns [N, roundup4(N)) are written as zeros.
 * transA=1 reduces over the long dimension: it runs split-K into `ws` (see
 * geogcn_gemm_workspace_bytes) and combines the slabs in fixed order (deterministic).
 *
 * `precision` selects how the products are formed (inputs, outputs and accumulation are fp32 in all):
 *   GEOGCN_GEMM_F32    v_mfma_f32_16x16x4_f32: an exact fp32 fma chain (the reference's sgemm class);
 *   GEOGCN_GEMM_BF16X3 each fp32 operand split exactly into three bf16 terms, six bf16 MFMA cross terms
 *                      per product: fp32-class accuracy (dropped terms O(2^-24 |a||b|); measured against fp64
 *                      1.0-1.4e-7 sum|a.b| at K = 300, the exact kernel 1.4-1.5e-7), 2.7x fewer MFMA cycles.  It is a
 *                      PERMISSION, not a promise: shapes no split-bf16 kernel takes run exact fp32 (csrc/gemm_x3.hip: whole
 *                      rows of A for M >= 4,096 (32,768 until round 6), K <= 1,024, N <= 1,024; transA = 1 in 128 / 160 x 256 / 320 tiles, several per
 *                      row of C beyond 320 columns; feature-panel outputs, geogcn_gemm_panels_f32, on the same whole-rows kernel
 *                      from 4,096 rows and on the staged split-bf16 kernel of csrc/gemm_bf16.hip below).  Non-finite operands: an Inf or NaN in A or B gives NaN in every output
 *                      it reaches (the split's residual Inf - Inf), where the exact kernels would give Inf for an Inf times a
 *                      non-zero; so does a finite |x| >= 0x1.ffp+127 (3.396e38: it rounds to the bf16 infinity).  Every other fp32
 *                      value, subnormals included, splits exactly;
 *   GEOGCN_GEMM_BF16   one bf16 term per operand (BASELINE config 5: "bf16 H.W on MFMA, fp32 accumulate").
 * BF16 applies to both orientations: transA = 1 (dW, the
 * reduction over the node dimension) rounds both operands to bf16 on their way into LDS for outputs wider than
 * 160 columns (narrower ones run F32); split-K slabs and their ordered combination stay fp32.            */
#define GEOGCN_GEMM_F32    0
#define GEOGCN_GEMM_BF16X3 1
#define GEOGCN_GEMM_BF16   2
size_t geogcn_gemm_workspace_bytes(int32_t transA, int32_t transB, int64_t M, int64_t N, int64_t K,
                                   int32_t precision);
int geogcn_gemm_f32(int32_t transA, int32_t transB, int64_t M, int64_t N, int64_t K,
                    const float* A, int64_t lda, const float* B, int64_t ldb,
                    float* C, int64_t ldc, const float* bias, int32_t act, int32_t accumulate,
                    int32_t precision, void* ws, size_t ws_bytes, void* stream);
/* bf16 configuration only: the GEOGCN_GEMM_BF16 product (transA = 0, no accumulate) with C stored as
 * bfloat16 (round to nearest even), the operand format of geogcn_spmm_csr_bf16b -- Z = H.W goes from the
 * MFMA accumulators to the SpMM without an fp32 round trip through HBM.  ldc is in ELEMENTS, % 8 == 0 and
 * >= roundup8(N); columns [N, roundup8(N)) are written as zeros.  Workspace as geogcn_gemm_workspace_bytes
 * (0, transB, M, N, K, GEOGCN_GEMM_BF16).                                                          */
int geogcn_gemm_f32_bf16c(int32_t transB, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
                          const float* B, int64_t ldb, uint16_t* C, int64_t ldc, const float* bias,
                          int32_t act, void* ws, size_t ws_bytes, void* stream);

/* The same product with C laid out as FEATURE PANELS: panels[q][i][j] = C[i][q * wp + j] for q < W, i < R, j < wp
 * (rows i >= M and columns >= N of the buffer are left untouched) -- the send buffer of the multi-GPU feature
 * repartition (one all-to-all around the SpMM) written straight from the MFMA accumulators instead
 * of by a separate pack pass.  transA = 0 only.  c_bf16 = 0: fp32 panels, wp % 4 == 0, any precision;  c_bf16 = 1:
 * bfloat16 panels (the SpMM operand and wire format of the bf16 configuration), wp % 8 == 0, precision must be
 * GEOGCN_GEMM_BF16.  No reference counterpart (the reference is single-device).                                */
int geogcn_gemm_panels_f32(int32_t transB, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B,
                           int64_t ldb, void* panels, int64_t R, int32_t W, int32_t wp, int32_t c_bf16, const float* bias,
                           int32_t act, int32_t precision, void* ws, size_t ws_bytes, void* stream);

/* Highway block, both weights in one launch (gcnmodel.py:281-286: the conv branch l_h and the gate l_t take the same
 * `incoming`).  Exact fp32 MFMA, same arithmetic per element as two geogcn_gemm_f32 calls:
 *   (C0, C1) = (act0(op(A) . B0 + bias0), act1(op(A) . B1 + bias1))       op(A) = A (transA = 0) or A^T (transA = 1)
 * transA = 0: the forward pair Z = H.Wh (raw, into the SpMM operand's pitch) and T = sigmoid(H.Wt + bt) -- every A
 * tile is multiplied by both weights inside one XCD (second read from L2).  transA = 1: dWh = H^T.dZ and
 * dWt = H^T.dU, one pass over H (split-K into `ws`, geogcn_gemm_dual_workspace_bytes; slabs combined in fixed order).
 * B0 is K x N0 (ldb0), B1 is K x N1 (ldb1).  If both act0 and act1 are non-linear they must be equal.
 * transA = 0 with a workspace of geogcn_gemm_dual_workspace_bytes: shapes like the GCN's (M >= 32,768 rows, K padding to 256 or
 * 304, N0 and N1 filling 320-column passes) run on the whole-rows kernel -- 64 rows of A per block read once, the weights
 * re-laid in fragment order into `ws` -- with bit-identical results; without it (ws too small / NULL) on the staged kernel.  */
/* `precision`: GEOGCN_GEMM_F32 (the exact kernels above) or GEOGCN_GEMM_BF16X3 (the same launches with fp32-class split-bf16
 * products where csrc/gemm_x3.hip takes the shape -- see geogcn_gemm_f32 -- exact fp32 otherwise); the workspace is sized for it.
 * GEOGCN_GEMM_BF16 is accepted with transA = 1 only (round 6; no bias / activation): both weight gradients of the bf16 configuration
 * in one launch of its A^T . B kernel where both segments are wider than 160 columns, its two launches otherwise (the forward pair of
 * that configuration is geogcn_gemm_dual_bf16 below).                                                                              */
size_t geogcn_gemm_dual_workspace_bytes(int32_t transA, int64_t M, int64_t N0, int64_t N1, int64_t K, int32_t precision);
int geogcn_gemm_dual_f32(int32_t transA, int64_t M, int64_t N0, int64_t N1, int64_t K, const float* A, int64_t lda,
                         const float* B0, int64_t ldb0, const float* B1, int64_t ldb1, float* C0, int64_t ldc0,
                         float* C1, int64_t ldc1, const float* bias0, int32_t act0, const float* bias1, int32_t act1,
                         int32_t precision, void* ws, size_t ws_bytes, void* stream);
/* The forward pair of the bf16 configuration (BASELINE configs[4]) in one launch: C0 = A . B0 stored as bfloat16 (c0_bf16 = 1: the
 * operand geogcn_spmm_csr_bf16b gathers; pitch a multiple of 8, pads as zeros) or fp32, C1 = act1(A . B1 + bias1) in fp32 --
 * bf16 products, fp32 accumulation, A read and rounded once (bf16 whole-rows kernel, widths <= 640 with K padding to 256 / 320 /
 * 608); other shapes run as the two separate geogcn_gemm_f32 / geogcn_gemm_f32_bf16c launches.  Bit-identical to those.        */
size_t geogcn_gemm_dual_bf16_workspace_bytes(int64_t M, int64_t N0, int64_t N1, int64_t K);
int geogcn_gemm_dual_bf16(int64_t M, int64_t N0, int64_t N1, int64_t K, const float* A, int64_t lda, const float* B0, int64_t ldb0,
                          const float* B1, int64_t ldb1, void* C0, int64_t ldc0, int32_t c0_bf16, float* C1, int64_t ldc1,
                          const float* bias1, int32_t act1, void* ws, size_t ws_bytes, void* stream);
/* Two products into one accumulator (what autodiff derives for the input of the highway block: dH = dZ.Wh^T + dU.Wt^T
 * [+ the carry gradient already in C]):  C[M x N] = A0 . op(B0) + A1 . op(B1) [+ C].  A0 is M x K0, A1 is M x K1;
 * transB = 1: B0 is N x K0, B1 is N x K1 (the weights as stored); transB = 0: B0 is K0 x N, B1 is K1 x N.
 * One pass over C instead of two accumulating calls.  `ws` (geogcn_gemm_kcat_workspace_bytes; may be NULL / 0): as for the
 * dual launch -- with it, eligible shapes run on the whole-rows kernel (same accumulation order, bit-identical).      */
size_t geogcn_gemm_kcat_workspace_bytes(int32_t transB, int64_t M, int64_t N, int64_t K0, int64_t K1, int32_t precision);
int geogcn_gemm_kcat_f32(int32_t transB, int64_t M, int64_t N, int64_t K0, int64_t K1, const float* A0, int64_t lda0,
                         const float* B0, int64_t ldb0, const float* A1, int64_t lda1, const float* B1, int64_t ldb1,
                         float* C, int64_t ldc, int32_t accumulate, int32_t precision, void* ws, size_t ws_bytes, void* stream);

/* The same with the block's CARRY gradient formed in the epilogue instead of being read from C:
 *   C[M x N] = A0 . op(B0) + A1 . op(B1) + G * (1 - T)          (G = gradient at the block's output, T = its gate; both M x N,
 * every operation rounded on its own: the value geogcn_highway_bwd_f32 would have stored as dHcarry) -- highway_bwd then
 * takes dHcarry = NULL and neither writes nor re-reads 4 M N bytes.  Bit-identical to geogcn_highway_bwd_f32 (with dHcarry) followed
 * by geogcn_gemm_kcat_f32(accumulate = 1).  Shapes the whole-rows kernel does not take (or ws too small) run as
 * geogcn_gate_carry_f32 into C + the accumulating call.  C must not alias G or T.                                            */
/* ... and, for the FIRST highway block, with the gradient of the layer below folded in as well:
 *   C = (A0 . op(B0) + A1 . op(B1) + G * (1 - T)) * (keep * scale) * (1 - Y^2)
 * i.e. dS0, the gradient at the pre-activation of the tanh layer whose dropped output feeds the block (gcnmodel.py:353,357:
 * dropout mask `keep` (bytes, pitch keepF, a multiple of 4) and 1/(1-p) = scale, Y = that layer's output) -- the pass
 * geogcn_act_bwd_f32 would make over dH.  Same bits as that pass (tested); the bias gradient with the bits of
 * geogcn_act_bwd_colsum_f32 is geogcn_colsum_rowblocks_f32 of C (its summation order), not geogcn_colsum_f32.                    */
int geogcn_gemm_kcat_gated_tanhbwd_f32(int32_t transB, int64_t M, int64_t N, int64_t K0, int64_t K1, const float* A0, int64_t lda0,
                                       const float* B0, int64_t ldb0, const float* A1, int64_t lda1, const float* B1, int64_t ldb1,
                                       float* C, int64_t ldc, const float* G, int64_t ldg, const float* T, int64_t ldt,
                                       const float* Y, int64_t ldy, const uint8_t* keep, int64_t keepF, float scale, int32_t precision,
                                       void* ws, size_t ws_bytes, void* stream);
/* (round 6) geogcn_gemm_kcat_f32 and geogcn_gemm_kcat_gated_f32 also take GEOGCN_GEMM_BF16: one launch of the bf16 whole-rows kernel
 * with both reductions' row tiles in LDS where K0 and K1 pad to the same 256 / 320 / 608 and N <= 640 (one fp32 accumulator over both
 * reductions: equal to the two launches to rounding, not bit for bit), the two launches otherwise; geogcn_gemm_kcat_workspace_bytes
 * covers both.  geogcn_gemm_kcat_gated_tanhbwd_f32 with GEOGCN_GEMM_BF16 applies the tanh gradient in a pass of its own.          */
/* ... and for ONE product: C[M x N] = A . op(B) + G * (1 - T) in any `precision` (the bf16 configuration formed dH_in with two
 * calls until round 6: this one, then an accumulating geogcn_gemm_f32).  ws as for geogcn_gemm_f32 (geogcn_gemm_workspace_bytes). */
int geogcn_gemm_gated_f32(int32_t transB, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B,
                          int64_t ldb, float* C, int64_t ldc, const float* G, int64_t ldg, const float* T, int64_t ldt,
                          int32_t precision, void* ws, size_t ws_bytes, void* stream);
int geogcn_gemm_kcat_gated_f32(int32_t transB, int64_t M, int64_t N, int64_t K0, int64_t K1, const float* A0, int64_t lda0,
                               const float* B0, int64_t ldb0, const float* A1, int64_t lda1, const float* B1, int64_t ldb1,
                               float* C, int64_t ldc, const float* G, int64_t ldg, const float* T, int64_t ldt, int32_t precision,
                               void* ws, size_t ws_bytes, void* stream);
/* ---- K7: fused Elemwise ------------------------------------------------------------------- */
/* Y = act(X + bias)                      gcnmodel.py:41-42,132-136 when not fused upstream      */
int geogcn_bias_act_f32(int64_t n, int32_t F, const float* X, int64_t ldx, const float* bias,
                        int32_t act, float* Y, int64_t ldy, void* stream);
/* MultiplicativeGatingLayer: Hout = T*Hc + (1-T)*H          gcnmodel.py:266                    */
int geogcn_highway_fwd_f32(int64_t n, int32_t F, const float* T, const float* Hc, const float* H,
                           int64_t ld, float* Hout, void* stream);
/* its gradient + tanh'/sigmoid' of the two branches (what autodiff derives for :266,:136,:286):
 *   dS = G*T*(1-Hc^2)   dU = G*(Hc-H)*T*(1-T)   dHcarry = G*(1-T)
 * dHcarry may be NULL (not stored: geogcn_gemm_kcat_gated_f32 forms it where it is consumed).   */
int geogcn_highway_bwd_f32(int64_t n, int32_t F, const float* G, const float* T, const float* Hc,
                           const float* H, int64_t ld, float* dS, int64_t ld_dS /* dS may use the line-aligned
                           pitch of an SpMM operand */, float* dU, float* dHcarry,
                           float* dbS /* nullable: column sums of dS = grad of the conv bias */,
                           float* dbU /* nullable: column sums of dU = grad of the gate bias */,
                           void* ws, size_t ws_bytes, void* stream);
/* column sums of X (n x F) taken in the order of the fused activation-gradient kernels (row blocks of
 * geogcn_highway_bwd_workspace_bytes' partition): the bias gradient of a dS that a product's epilogue wrote, with the bits
 * geogcn_act_bwd_colsum_f32 would have given it.  ws: geogcn_highway_bwd_workspace_bytes(n, F).                       */
int geogcn_colsum_rowblocks_f32(int64_t n, int32_t F, const float* X, int64_t ldx, float* out, void* ws, size_t ws_bytes,
                                void* stream);
/* out = G * (1 - T): the carry gradient alone (the arithmetic of dHcarry above)                 */
int geogcn_gate_carry_f32(int64_t n, int32_t F, const float* G, int64_t ldg, const float* T, int64_t ldt, float* out,
                          int64_t ldo, void* stream);
/* the same with dS stored as bfloat16 (round to nearest even; ld_dS16 in elements, a multiple of 8, the whole pitch written,
 * pads as zeros): in the bf16 configuration dS is only ever gathered by A^T . dS (geogcn_spmm_csr_bf16b), so the fp32 copy
 * and the geogcn_cast_bf16_f32 pass over it are not needed.  dbS is still the column sum of the fp32 values.  F <= 1024.     */
int geogcn_highway_bwd_bf16s_f32(int64_t n, int32_t F, const float* G, const float* T, const float* Hc,
                                 const float* H, int64_t ld, uint16_t* dS16, int64_t ld_dS16, float* dU, float* dHcarry,
                                 float* dbS, float* dbU, void* ws, size_t ws_bytes, void* stream);
/* workspace of the fused column sums (deterministic two-pass); 0 when they are not requested */
size_t geogcn_highway_bwd_workspace_bytes(int64_t n, int32_t F);
/* dS = G [* keep_mask * scale] * act'(Y) with act' expressed through the layer OUTPUT Y:
 * tanh: 1 - Y^2 (gcnmodel.py:42,136), sigmoid: Y(1-Y) (gcnmodel.py:286), none: 1.  The optional
 * mask folds in the dropout that follows layer 0 (gcnmodel.py:357); mask may be NULL.           */
int geogcn_act_bwd_f32(int64_t n, int32_t F, const float* G, const float* Y, int64_t ld, int32_t act,
                       const uint8_t* keep_mask, float scale, float* dS, int64_t ld_dS, void* stream);
/* the same with the bias gradient db[F] = column sums of dS in the same pass (deterministic: fixed block / thread
 * partition, partials combined in order); workspace: geogcn_highway_bwd_workspace_bytes(n, F) suffices       */
int geogcn_act_bwd_colsum_f32(int64_t n, int32_t F, const float* G, const float* Y, int64_t ld, int32_t act,
                              const uint8_t* keep_mask, float scale, float* dS, int64_t ld_dS, float* db,
                              void* ws, size_t ws_bytes, void* stream);
/* Y += X over n_floats contiguous floats (gradient accumulation where a layer output feeds
 * several consumers: Theano's Elemwise{add} in the autodiff graph of gcnmodel.py:266,288)      */
int geogcn_add_inplace_f32(int64_t n_floats, const float* X, float* Y, void* stream);
/* out[F] = sum over rows of X (bias gradients: Sum{axis=0}); deterministic two-pass            */
size_t geogcn_colsum_workspace_bytes(int64_t n, int32_t F);
int geogcn_colsum_f32(int64_t n, int32_t F, const float* X, int64_t ldx, float* out,
                      void* ws, size_t ws_bytes, void* stream);

/* ---- K10: lasagne DropoutLayer (gcnmodel.py:357) -----------------------------------------
 * mask generation is counter-based Philox4x32-10 (Theano's MRG31k3p stream cannot be matched;
 * parity runs inject the mask).  keep_mask[i*F + j] in {0,1}, dense (no pitch).                */
int geogcn_dropout_mask_philox(int64_t n, int32_t F, float p_drop, uint64_t seed, uint64_t offset,
                               uint8_t* keep_mask, void* stream);
/* Captured steps (hipGraph): a replay cannot change kernel arguments, so the stream position comes from a
 * device-resident call counter: offset = (calls_dev[0] * per_call_elems + base_elems) / 4 quads -- the value
 * the host would have passed to geogcn_dropout_mask_philox.  geogcn_counter_add_i64 advances such a counter
 * from inside the captured stream.                                                                  */
int geogcn_dropout_mask_philox_ctr(int64_t n, int32_t F, float p_drop, uint64_t seed, const int64_t* calls_dev,
                                   int64_t per_call_elems, int64_t base_elems, uint8_t* keep_mask, void* stream);
int geogcn_counter_add_i64(int64_t* counter_dev, int64_t delta, void* stream);
/* Dropout on the VALUES of a sparse matrix (SparseInputDropoutLayer, gcnmodel.py:44-70): element (i, j) of the
 * logical n_rows_logical x n_cols_logical matrix is kept (and scaled by 1/(1-p)) iff the Philox draw keyed by its
 * position i * n_cols_logical + j (stream `call`) is below 1-p -- the same decision in every device layout of the
 * matrix: its CSR (transposed = 0), the CSR of (a part of) its transpose (transposed = 1: row = j, colidx = i) and a
 * dense panel of selected columns (head_idx[k] = j).                                                          */
int geogcn_dropout_csr_f32(int64_t n_rows, const int32_t* rowptr, const int32_t* colidx, const float* val_in,
                           float* val_out, int64_t n_rows_logical, int64_t n_cols_logical, int32_t transposed,
                           float p_drop, uint64_t seed, uint64_t call, void* stream);
int geogcn_dropout_panel_f32(int64_t n, int32_t K, const float* panel_in, int64_t ld, const int32_t* head_idx,
                             int64_t n_cols_logical, float p_drop, uint64_t seed, uint64_t call, float* panel_out,
                             void* stream);
/* Y = X * keep_mask / (1-p)  (forward; the backward is the same call on the gradient)          */
int geogcn_dropout_apply_f32(int64_t n, int32_t F, const float* X, int64_t ld,
                             const uint8_t* keep_mask, float p_drop, float* Y, void* stream);

/* ---- K8/K9: Softmax, categorical_crossentropy, argmax, row gather / scatter -----------------
 * probs = softmax_rows(logits) (max-subtracted), argmax = first index of the row maximum
 * (gcnmodel.py:374,377,379,394); argmax_out may be NULL.                                        */
int geogcn_softmax_rows_f32(int64_t n, int32_t C, const float* logits, int64_t ldl, float* probs,
                            int64_t ldp, int32_t* argmax_out, void* stream);
/* sums over the gathered rows idx[0..n_idx): loss_sum = sum -log P[idx[j], y[j]],
 * correct = #{argmax P[idx[j]] == y[j]}   (gcnmodel.py:376-382,389); out2 = {loss_sum, correct} */
size_t geogcn_ce_metrics_workspace_bytes(int64_t n_idx);
int geogcn_ce_metrics_f32(int32_t C, const float* probs, int64_t ldp, const int32_t* argmax /*nullable:
                          per-row argmax from geogcn_softmax_rows_f32*/, const int32_t* idx,
                          int64_t n_idx, const int32_t* y, float* out2, void* ws, size_t ws_bytes,
                          void* stream);
/* dlogits = 0; dlogits[idx[j], :] += (P[idx[j], :] - onehot(y[j])) * inv_n  (autodiff of :376,:382;
 * inv_n = 1/len(train_indices) -- passed explicitly because a row-partitioned rank holds only
 * its share of the indices while the mean runs over all of them)                                */
int geogcn_softmax_ce_bwd_f32(int64_t n, int32_t C, const float* probs, int64_t ldp,
                              const int32_t* idx, int64_t n_idx, const int32_t* y, float inv_n,
                              float* dlogits, int64_t ldd, void* stream);
/* the same plus db[C] = column sums of dlogits (the output layer's bias gradient, gcnmodel.py:155-157 autodiff)
 * computed from the indexed rows while they are scattered -- no pass over the N x C matrix.  C <= 1024.     */
size_t geogcn_softmax_ce_bwd_db_workspace_bytes(int32_t C);
int geogcn_softmax_ce_bwd_db_f32(int64_t n, int32_t C, const float* probs, int64_t ldp, const int32_t* idx,
                                 int64_t n_idx, const int32_t* y, float inv_n, float* dlogits, int64_t ldd,
                                 float* db, void* ws, size_t ws_bytes, void* stream);
/* COMPACT form: drows[j, :] = (P[idx[j], :] - onehot(y[j])) * inv_n for j < n_idx -- one row per INDEX instead of a
 * zero-filled N x C matrix with the indexed rows scattered into it -- plus db as above.  The consumer multiplies by the
 * transposed adjacency restricted to the indexed COLUMNS, renumbered by position in idx (the structural zeros of the
 * N x C gradient are never stored, written or gathered).  ldd >= roundup4(C); pad columns [C, roundup4(C)) are zeroed. */
int geogcn_softmax_ce_rows_bwd_db_f32(int32_t C, const float* probs, int64_t ldp, const int32_t* idx, int64_t n_idx,
                                      const int32_t* y, float inv_n, float* drows, int64_t ldd, float* db, void* ws,
                                      size_t ws_bytes, void* stream);
/* out[j, :] = X[idx[j], :]   AdvancedSubtensor1 (gcnmodel.py:376,378,393); dense out (pitch F)  */
int geogcn_gather_rows_f32(int32_t F, const float* X, int64_t ldx, const int32_t* idx, int64_t n_idx,
                           float* out, int64_t ldo, void* stream);

/* out[idx[j], :] = src[j, :]  (rows of a small result placed into a larger matrix; idx unique)   */
int geogcn_scatter_rows_f32(int32_t F, const float* src, int64_t lds, const int32_t* idx, int64_t n_idx,
                            float* out, int64_t ldo, void* stream);

/* ---- multi-GPU repartition helpers (no reference counterpart: the reference is single-device) -------
 * Row-partitioned activations <-> feature-partitioned SpMM operands, exchanged with an all-to-all.
 * pack:   out[q][i][j] = X[i][q*wp + j]   for q < W, i < R, j < wp   (0 where i >= n_rows or col >= F)
 * unpack: Y[i][q*wp + j] = in[q][i][j]    for i < n_rows, col < F    (pad columns of Y up to roundup4(F) = 0)
 * `out` / `in` are dense [W][R][wp] fp32 arrays, wp % 4 == 0.                                        */
int geogcn_pack_panels_f32(int64_t n_rows, int64_t R, int32_t F, const float* X, int64_t ldx, int32_t W,
                           int32_t wp, float* out, void* stream);
int geogcn_unpack_panels_f32(int64_t n_rows, int64_t R, int32_t F, const float* in, int32_t W, int32_t wp,
                             float* Y, int64_t ldy, void* stream);

/* ---- collectives of the partitioned step: RCCL over xGMI, one process per GPU ------------------------------------
 * No reference counterpart (the reference is one process on one device; the step partitioned here is
 * gcnmodel.py:409-430).  These are what a host WITHOUT torch.distributed binds; RCCL is looked up at run time (dlopen),
 * so the library loads without it and geogcn_comm_available() says whether the calls below can work.
 * Protocol: rank 0 calls geogcn_comm_unique_id and hands the 128 bytes to every rank by any means (file, socket, MPI);
 * every rank selects its GPU (hipSetDevice) and calls geogcn_comm_init_rank -- collective, blocks until all have
 * joined.  All operations are enqueued on `stream` and return immediately; buffers are borrowed until the stream has
 * passed them.  Errors: <0 GEOGCN_E_*, >0 a hipError_t, >= 1000 an ncclResult_t + 1000 (text in geogcn_last_error).  */
#define GEOGCN_COMM_ID_BYTES 128
typedef struct geogcn_comm geogcn_comm;
int     geogcn_comm_available(void);
int     geogcn_comm_unique_id(void* id_out, size_t id_bytes);
int     geogcn_comm_init_rank(const void* id, int32_t world, int32_t rank, geogcn_comm** out);
void    geogcn_comm_destroy(geogcn_comm* comm);
int32_t geogcn_comm_world(const geogcn_comm* comm);
int32_t geogcn_comm_rank(const geogcn_comm* comm);
/* gradients, loss sums:  buf[i] = sum over ranks of buf[i]  (in place; the summation order is RCCL's, hence the
 * 1e-6 relative tolerance of the N-GPU == 1-GPU tests on everything that passes through it)                        */
int     geogcn_comm_allreduce_sum_f32(geogcn_comm* comm, float* buf, int64_t n, void* stream);
/* row shards of Z / dS:  recv = [rank 0's bytes | rank 1's | ...], bytes_per_rank each (equal slots: the cost-balanced
 * split pads the shorter shards, dist.py RowPartition.R); in place when send == recv + rank * bytes_per_rank        */
int     geogcn_comm_allgather(geogcn_comm* comm, const void* send, void* recv, int64_t bytes_per_rank, void* stream);
/* feature repartition:  panel p of `send` goes to rank p, panel q of `recv` comes from rank q (the [W][R][wp] layout
 * geogcn_gemm_panels_f32 writes and the narrow SpMM reads); W - 1 point-to-point transfers in one group, one per xGMI
 * link.  send and recv must not alias.                                                                              */
int     geogcn_comm_alltoall(geogcn_comm* comm, const void* send, void* recv, int64_t bytes_per_peer, void* stream);
/* halo exchange (graphs WITH locality, dist.py HaloPlan): rank p receives send_bytes[p] bytes of `send` -- the rows of this
 * rank's block that p's rows of A_hat reference, packed back to back in peer order -- and recv_bytes[p] bytes from p land in
 * `recv`, again back to back in peer order.  Both vectors are HOST arrays of `world` entries; zero-sized pieces are skipped
 * (the lists are symmetric: what I do not send to p, p does not post a receive for).  send and recv must not alias.        */
int     geogcn_comm_alltoallv(geogcn_comm* comm, const void* send, const int64_t* send_bytes, void* recv,
                              const int64_t* recv_bytes, void* stream);

/* ---- K11/K12: lasagne.updates.adam (+ l1/l2 penalty gradient), gcnmodel.py:383-387,407 ------
 * flat arenas of n floats: t is the step index AFTER increment (1 for the first call).
 *   g' = g + regmask*(l1*sign(p) + 2*l2*p);  m = b1 m + (1-b1) g';  v = b2 v + (1-b2) g'^2
 *   p -= lr*sqrt(1-b2^t)/(1-b1^t) * m / (sqrt(v) + eps)
 * regmask (nullable) is 1.0 on weight matrices ("regularizable") and 0 on biases / padding.     */
int geogcn_adam_step_f32(int64_t n, float* p, float* g /* updated to g' when l1|l2 != 0 */, float* m, float* v,
                         const float* regmask, float lr, float b1, float b2, float eps, int32_t t,
                         float l1, float l2, void* stream);
/* Same update with the step index on the device (captured steps): state_dev[0] = number of steps taken so
 * far (incremented by this call before use), state_dev[1] = scratch (bits of the fp32 step size a_t).   */
int geogcn_adam_step_ctr_f32(int64_t n, float* p, float* g, float* m, float* v, const float* regmask, float lr,
                             float b1, float b2, float eps, int64_t* state_dev, float l1, float l2, void* stream);
/* penalty value: sum regmask*(l1*|p| + l2*p^2) -> out[0] (deterministic two-pass)               */
int geogcn_reg_penalty_f32(int64_t n, const float* p, const float* regmask, float l1, float l2,
                           float* out, void* ws, size_t ws_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GEOGCN_H */
