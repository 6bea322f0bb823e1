/*
 * mggcn.h -- C ABI of libmggcn_hip.so, the MI355X (gfx950) engine behind the
 * MG-GCN hot path: CSR SpMM aggregation (A.H forward, A^T.dH backward), the
 * dense H.W transform and the element-wise / loss / optimiser kernels of one
 * training epoch, plus the host-side graph preprocessing that feeds them.
 *
 * The reference has no FFI: it is a header-only C++ template library whose device
 * code is reached through forward-declared launchers
 *     void f(cudaStream_t, T* ..., size_t size, size_t m, ...)
 * (src/cuda_utils.hpp:398-468  <->  src/cuda_utils.cu:229-451) and through
 * cusparseSpMM / cublasSgemm handles bound to a stream (src/matrix.hpp:86-87).
 * That host-header <-> device-TU seam is the boundary this file replaces: every
 * entry point below names the reference interface it stands in for.
 *
 * Conventions (same as the reference):
 *  - plain pointers and sizes; no HIP, torch or C++ types.  A stream is an opaque
 *    void* (a hipStream_t; NULL = the device's default stream).  torch users pass
 *    torch.cuda.current_stream().cuda_stream.
 *  - enqueue-only: nothing here synchronises or allocates on the launch path
 *    (the *_create / *_malloc / *_host functions are the exceptions, by name).
 *  - the callee never takes ownership of a buffer.
 *  - fail-fast: a HIP error or a violated precondition prints
 *    "MGGCN ... failed at file:line" and calls exit(EXIT_FAILURE), exactly like
 *    CHECK_CUDA / CHECK_CUSPARSE (src/mg_gcn.hpp:31-68).  No status codes.
 *  - dense matrices are row-major fp32 with an explicit leading dimension
 *    (CUSPARSE_ORDER_ROW, src/matrix.hpp:508); CSR is u32 indptr / u32 indices /
 *    f32 values, zero-based (src/matrix.hpp:217-221, :271).
 *  - "size" is the element count n*m and "m" the row width, as in the launchers.
 *
 * All citations are relative to the reference tree (GT-TDAlab/MG-GCN).
 */
#ifndef MGGCN_H_
#define MGGCN_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MGGCN_ABI_VERSION 1

typedef void *mggcn_stream_t; /* hipStream_t */
typedef void *mggcn_event_t;  /* hipEvent_t  */

/* ======================================================================== *
 * Runtime: the per-GPU `context` (src/matrix.hpp:69-158)
 * ======================================================================== */
int mggcn_abi_version(void);
/* cudaGetDeviceCount / cudaSetDevice (context::set, src/matrix.hpp:90-92) */
int mggcn_device_count(void);
void mggcn_set_device(int device);
int mggcn_get_device(void);
/* context::sync = cudaDeviceSynchronize (src/matrix.hpp:94-97) */
void mggcn_device_synchronize(void);
/* stream_create(i, priority) (src/matrix.hpp:53-60): high_priority != 0 asks for the
 * highest priority the device offers (the reference's comm stream), 0 for the lowest
 * (its compute stream). */
mggcn_stream_t mggcn_stream_create(int high_priority);
void mggcn_stream_destroy(mggcn_stream_t stream);
/* Frees the small per-(device, stream) reduction scratch that mggcn_abssum_f32 / the fused loss allocate on first use
 * (synchronises that stream).  mggcn_stream_destroy calls it; a host layer that brings its own streams (e.g. torch's)
 * calls it when it drops one. */
void mggcn_stream_release_scratch(mggcn_stream_t stream);
void mggcn_stream_synchronize(mggcn_stream_t stream);
/* event_create / context::record / context::wait / context::measure
 * (src/matrix.hpp:62-67, :107-117, :138-144) */
mggcn_event_t mggcn_event_create(void);
void mggcn_event_destroy(mggcn_event_t event);
void mggcn_event_record(mggcn_event_t event, mggcn_stream_t stream);
void mggcn_stream_wait_event(mggcn_stream_t stream, mggcn_event_t event);
void mggcn_event_synchronize(mggcn_event_t event);
float mggcn_event_elapsed_ms(mggcn_event_t begin, mggcn_event_t end);
/* cuda_malloc / cudaFree deleter (src/mg_gcn.hpp:84-90).  The reference's
 * cudaMallocManaged buffers (src/mg_gcn.hpp:74-82) become device memory plus an
 * explicit pinned host mirror: managed memory needs XNACK, which this pool lacks. */
void *mggcn_malloc(size_t bytes);
void mggcn_free(void *device_ptr);
void *mggcn_malloc_host(size_t bytes);
void mggcn_free_host(void *host_ptr);
/* cudaMemcpyAsync / cudaMemsetAsync (dn_matrix::copy_to, ::zero, src/matrix.hpp:547-566) */
void mggcn_memcpy_h2d(void *dst, const void *src, size_t bytes, mggcn_stream_t stream);
void mggcn_memcpy_d2h(void *dst, const void *src, size_t bytes, mggcn_stream_t stream);
void mggcn_memcpy_d2d(void *dst, const void *src, size_t bytes, mggcn_stream_t stream);
void mggcn_memset_zero(void *dst, size_t bytes, mggcn_stream_t stream);

/* ======================================================================== *
 * SpMM  C = alpha * A * B + beta * C     (the hot kernel)
 * replaces matmul(context, csr_matrix, dn_matrix B, dn_matrix C, ext_buffer,
 * alpha, beta, alg) = cusparseSpMM op N/N (src/cuda_utils.hpp:15-32) and its
 * workspace query get_matmul_buffer (src/cuda_utils.hpp:94-102).
 * ======================================================================== */

/* The "external buffer" of the reference becomes a plan, built once per matrix from
 * the HOST copy of the CSR arrays; it holds only device memory it allocated itself:
 *  - row-split metadata that load-balances heavy-tailed degree distributions (rows
 *    longer than the split threshold are cut into work items whose partial sums are
 *    combined in a fixed order -> bitwise reproducible) and the partial-sum workspace
 *    for feature widths up to max_d;
 *  - when host_indices / host_values are given and the matrix is large (>= 2^20
 *    non-zeros), the column-panel "sweep" form: the non-zeros re-cut into equal-work
 *    per-wave streams sorted by column panel, so that every wave on the chip walks B
 *    in the same order and the active panel stays L2-resident (a private device copy
 *    of the matrix, 8 B per non-zero).  Pass NULL for both to skip it.
 * The arrays later passed to mggcn_spmm_csr_f32 with a plan must hold the matrix the
 * plan was built for (the sweep form reads its own copy).
 * No input order is assumed: rows may or may not be sorted by column (heavy rows are cut into interleaved slices), and
 * a vertex order with locality (an unpermuted community graph: most non-zeros near the diagonal) is detected at plan
 * time -- the sweep form then works on a fixed pseudo-random relabelling of the columns and copies B into a plan-owned
 * scratch in that order at the start of every call (the caller's B is never written).
 * Re-entrancy: every entry point of this header is enqueue-only and may be called from any stream, but a
 * PLAN owns mutable device scratch (partial-sum slots of sliced rows, the re-pitched copy of B of the narrow
 * form): one plan may be in flight on ONE stream at a time.  Calls on the same stream are ordered and safe;
 * to multiply by the same matrix on two streams concurrently, build two plans.  (mggcn_abssum_f32 keeps its
 * reduction scratch per stream: concurrent sums on different streams are safe.)
 * Plan CREATION is re-entrant: several host threads may build plans at the same time (each thread makes its target
 * device current first, mggcn_set_device) -- the host layers build a model's four plans side by side that way. */
typedef struct mggcn_spmm_plan mggcn_spmm_plan;

mggcn_spmm_plan *mggcn_spmm_plan_create(uint32_t n_rows, uint32_t n_cols,
                                        const uint32_t *host_indptr, const uint32_t *host_indices,
                                        const float *host_values, uint32_t max_d);
/* Same, with the feature width the plan will mostly be used at (the reference sizes its cuSPARSE
 * workspace per (A, B, C) triple as well, src/cuda_utils.hpp:94-102, and caches it per width,
 * src/gcn.hpp:28-31).  d_hint in 1..64 builds the sweep form for NARROW rows (four entries per
 * gather instruction, B re-pitched to 16-byte rows in a plan-owned scratch); 0 or > 64 is the
 * form mggcn_spmm_plan_create builds.  Any plan serves any width <= max_d; the hint only picks
 * which one is fast. */
mggcn_spmm_plan *mggcn_spmm_plan_create_for(uint32_t n_rows, uint32_t n_cols,
                                            const uint32_t *host_indptr, const uint32_t *host_indices,
                                            const float *host_values, uint32_t max_d, uint32_t d_hint);
void mggcn_spmm_plan_destroy(mggcn_spmm_plan *plan);
/* A caller that builds n plans side by side (one host thread each) says so before it starts: every builder then threads
 * its own passes over cores / n instead of all the cores (MGGCN_HOST_THREADS in the environment still wins).  Set it back
 * to 1 afterwards.  Process-wide. */
void mggcn_spmm_plan_concurrent_builders(uint32_t n);
/* Plans built from now on leave at least n compute units' worth of wave slots free in EVERY launch round: for SpMMs that run
 * while a collective kernel (RCCL) shares the device -- a round of exactly the resident set would end in a second, nearly empty
 * pass for the workgroups the foreign kernel displaced (+57 % measured).  A minimum, not a cut: a matrix
 * whose tasks leave that room anyway (one round, not full) keeps the full round size; only plans whose rounds would be full are
 * built on smaller rounds (fewer waves in flight: -5 % when nobody shares the device).  The distributed host layers set 12
 * around their plan builds and 0 afterwards; MGGCN_SPMM_RESERVED_CUS in the environment overrides.  Process-wide. */
void mggcn_spmm_plan_reserved_cus(uint32_t n);
/* introspection (tests, DESIGN.md figures) */
uint32_t mggcn_spmm_plan_num_items(const mggcn_spmm_plan *plan);
uint32_t mggcn_spmm_plan_num_split_rows(const mggcn_spmm_plan *plan);
uint32_t mggcn_spmm_plan_num_sweep_tasks(const mggcn_spmm_plan *plan); /* 0: no sweep form */
uint32_t mggcn_spmm_plan_num_launches(const mggcn_spmm_plan *plan, uint32_t d); /* kernel launches per SpMM call at width d */
size_t mggcn_spmm_plan_bytes(const mggcn_spmm_plan *plan);
uint32_t mggcn_spmm_plan_num_slices(const mggcn_spmm_plan *plan);      /* column slices of the sweep form (0: none) */
/* One line of text with what the plan builder measured and decided: form (rowsplit / sweep / sweep-narrow), share of the
 * non-zeros in the 1 % most popular columns and the flag derived from it, mean (panel,row) run length, column slices,
 * device bytes, host build seconds, then per slice: tasks, launch rounds, panel rows, run padding, lanes per entry,
 * padded entry count.  snprintf semantics (returns the length written).  MGGCN_SPMM_PLAN_LOG=1 in the environment
 * prints the same line to stderr whenever a plan is created.  The tuning knobs (MGGCN_SPMM_*) are read once, here,
 * never on the launch path. */
int mggcn_spmm_plan_describe(const mggcn_spmm_plan *plan, char *out, size_t cap);
/* diagnostics (experiments only): n_blocks workgroups of 256 threads that hold their wave slots until *stop_flag (device memory, or mapped
 * pinned host memory) is non-zero or max_microseconds have passed -- a stand-in for the channels of a collective
 * kernel sharing the device with the SpMM (profiles/experiments/coresident_r04.py). */
void mggcn_debug_occupy_cus(mggcn_stream_t stream, uint32_t n_blocks, uint32_t max_microseconds, const uint32_t *stop_flag);
/* diagnostics: with MGGCN_SPMM_STAMPS=1 in the environment at plan creation the d >= 96 sweep kernel records, per
 * one-wave task of column slice `slice`, {start, end} on the 100 MHz constant clock and {HW_ID << 32 | blockIdx << 4 | XCC id} of
 * its LAST launch; this copies them out (3 x u64 per task, blocking) and returns the task count.  Never set in a
 * timed run. */
uint32_t mggcn_spmm_plan_read_stamps(const mggcn_spmm_plan *plan, uint32_t slice, uint64_t *host_out,
                                     uint32_t capacity_tasks);

/* flags */
#define MGGCN_SPMM_DEFAULT 0u
/* fused epilogue: C = leaky_relu(alpha*A*B + beta*C, slope) -- folds the
 * leaky_relu_forward launch of gcn_layer::operator() (src/gcn.hpp:447-452) */
#define MGGCN_SPMM_LEAKY_RELU 1u

/* A: n_rows x n_cols CSR (device pointers), B: n_cols x d (ldb >= d), C: n_rows x d
 * (ldc >= d).  beta == 0 never reads C.  C must not alias B.  plan may be NULL
 * (one wave per row in row order: correct for any input, slow on skewed degrees).
 * slope is only read with MGGCN_SPMM_LEAKY_RELU. */
void mggcn_spmm_csr_f32(mggcn_stream_t stream, const mggcn_spmm_plan *plan, uint32_t n_rows,
                        uint32_t n_cols, const uint32_t *indptr, const uint32_t *indices,
                        const float *values, const float *B, size_t ldb, float *C, size_t ldc,
                        uint32_t d, float alpha, float beta, uint32_t flags, float slope);

/* ======================================================================== *
 * Dense GEMM  C = alpha * op(A) * op(B) + beta * C, row-major  (the MFMA path)
 * replaces matmul(context, dn A, dn B, dn C, alpha, beta, A_T, B_T) =
 * cublasSgemm with swapped operands (src/cuda_utils.hpp:149-172).
 * op(A) is M x K, op(B) is K x N; lda/ldb/ldc are the STORED leading dimensions.
 * fp32 in, fp32 accumulate on v_mfma_f32_32x32x2_f32 (exact fp32 products).
 * Tall reductions (K >> M*N, e.g. G_W = X^T G) are split over K; the partial
 * sums are combined in a fixed order inside `workspace` (bitwise reproducible).
 * workspace may be NULL when mggcn_gemm_workspace_bytes(...) == 0.
 * ======================================================================== */
size_t mggcn_gemm_workspace_bytes(int trans_a, int trans_b, uint32_t M, uint32_t N, uint32_t K);
void mggcn_gemm_f32(mggcn_stream_t stream, int trans_a, int trans_b, uint32_t M, uint32_t N,
                    uint32_t K, float alpha, const float *A, size_t lda, const float *B, size_t ldb,
                    float beta, float *C, size_t ldc, void *workspace, size_t workspace_bytes);
/* C = alpha * op(A) * op(B) + 1 * bias^T   (bias: N floats added to every row of C).
 * The reference's linear forward is broadcast_rows(b -> XW) followed by an sgemm with beta = 1
 * (src/gcn.hpp:116-123); this is the same sum in one pass: no broadcast kernel, no read of C.
 * Workspace as for mggcn_gemm_f32. */
void mggcn_gemm_bias_f32(mggcn_stream_t stream, int trans_a, int trans_b, uint32_t M, uint32_t N,
                         uint32_t K, float alpha, const float *A, size_t lda, const float *B, size_t ldb,
                         const float *bias, float *C, size_t ldc, void *workspace, size_t workspace_bytes);
/* C = alpha * A^T * B  AND  colsum = alpha * 1^T B  in one pass (A stored [K x M], B stored [K x N]).
 * The reference's linear::backward runs G_b = 1^T G as an sgemm with a ones vector and then G_W = X^T G
 * (src/gcn.hpp:125-134): two passes over G.  Here the workgroups of the first M-tile add up the B tiles they stage for
 * the MFMAs anyway -- no second read of G, no extra launch.  Split-K slabs carry the sums as one extra row; combined in
 * the same fixed order (reproducible).  Workspace: mggcn_gemm_tn_colsum_workspace_bytes. */
size_t mggcn_gemm_tn_colsum_workspace_bytes(uint32_t M, uint32_t N, uint32_t K);
void mggcn_gemm_tn_colsum_f32(mggcn_stream_t stream, uint32_t M, uint32_t N, uint32_t K, float alpha, const float *A,
                              size_t lda, const float *B, size_t ldb, float *C, size_t ldc, float *colsum,
                              void *workspace, size_t workspace_bytes);
/* C = (alpha * op(A) * op(B)) .* (Z > 0 ? 1 : slope)   (Z: M x N, ldz >= N; C is never read).
 * Folds the NEXT leaky_relu_backward launch into the GEMM that produces its gradient operand: the reference
 * runs G_out = G . W^T in layer i+1 (src/gcn.hpp:135-137) and then, in layer i, leaky_relu_backward(Z_i, G_out)
 * (src/gcn.hpp:462-468, src/cuda_utils.cu:33-38) -- a pass that reads Z_i and G_out and writes T.  Z_i is layer
 * i+1's own input X, so the mask is applied on the accumulator tile in this GEMM's epilogue: one read of Z, no
 * extra pass, no extra write.  Same products and the same single multiply as the two launches -> bitwise equal. */
void mggcn_gemm_lrelu_bwd_f32(mggcn_stream_t stream, int trans_a, int trans_b, uint32_t M, uint32_t N,
                              uint32_t K, float alpha, const float *A, size_t lda, const float *B, size_t ldb,
                              const float *Z, size_t ldz, float slope, float *C, size_t ldc, void *workspace,
                              size_t workspace_bytes);

/* ======================================================================== *
 * Element-wise / row kernels: one entry point per live launcher of
 * src/cuda_utils.cu (declared src/cuda_utils.hpp:398-468).  Same argument
 * order as the launcher it replaces, stream first.  In-place use (out == in)
 * is legal wherever the reference uses it (src/gcn.hpp:449, :464).
 * ======================================================================== */
/* leaky_relu_forward  (src/cuda_utils.cu:26-31, :229-234)  out = max(in, alpha*in) */
void mggcn_leaky_relu_forward_f32(mggcn_stream_t stream, const float *in, float *out, size_t size,
                                  float alpha);
/* leaky_relu_backward (src/cuda_utils.cu:33-38, :236-241)  G_out = in > 0 ? G_in : alpha*G_in */
void mggcn_leaky_relu_backward_f32(mggcn_stream_t stream, const float *in, const float *G_in,
                                   float *G_out, size_t size, float alpha);
/* broadcast_rows (src/cuda_utils.cu:40-51, :243-248)  mat[i,:] = row (discard) or += row */
void mggcn_broadcast_rows_f32(mggcn_stream_t stream, const float *row, float *mat, size_t size,
                              size_t m, int discard);
/* scale_rows (src/cuda_utils.cu:75-79, :264-269)  mat[i,:] /= scalar[i] */
void mggcn_scale_rows_f32(mggcn_stream_t stream, float *mat, const float *scalar, size_t size,
                          size_t m);
/* max_rows (src/cuda_utils.cu:95-104, :271-276) */
void mggcn_max_rows_f32(mggcn_stream_t stream, const float *mat, float *maxs, size_t size, size_t m);
/* max_row_indices (src/cuda_utils.cu:119-133, :285-290): first maximum wins */
void mggcn_max_row_indices_f32(mggcn_stream_t stream, const float *mat, int32_t *maxs, size_t size,
                               size_t m);
/* index_log_rows (src/cuda_utils.cu:142-150, :299-304)  values[i] = log(mat[i, indices[i]]) */
void mggcn_index_log_rows_f32(mggcn_stream_t stream, const float *mat, const int32_t *indices,
                              float *values, size_t size, size_t m);
/* add_indexed_rows (src/cuda_utils.cu:159-164, :313-319)  mat[i, indices[i]] += alpha */
void mggcn_add_indexed_rows_f32(mggcn_stream_t stream, float *mat, const int32_t *indices,
                                float alpha, size_t size, size_t m);
/* is_equal (src/cuda_utils.cu:180-184, :336-341)  out[i] = (a[i] == b[i]) */
void mggcn_is_equal_i32(mggcn_stream_t stream, const int32_t *a, const int32_t *b, float *out,
                        size_t size);
/* subtract_rows_exp (src/cuda_utils.cu:192-200, :350-355)  out = exp(mat - scalar[row]) */
void mggcn_subtract_rows_exp_f32(mggcn_stream_t stream, const float *mat, const float *scalar,
                                 float *out, size_t size, size_t m);
/* axpby (src/cuda_utils.cu:81-86, :364-369)   B = alpha*A + beta*B */
void mggcn_axpby_f32(mggcn_stream_t stream, const float *A, float *B, float alpha, float beta,
                     size_t size);
/* aaxpby (src/cuda_utils.cu:88-93, :371-376)  B = alpha*A*A + beta*B */
void mggcn_aaxpby_f32(mggcn_stream_t stream, const float *A, float *B, float alpha, float beta,
                      size_t size);
/* adam_final (src/cuda_utils.cu:208-218, :378-383)  p -= (lr/c1) * m / (sqrt(v/c2) + eps) */
void mggcn_adam_final_f32(mggcn_stream_t stream, float *param, const float *m, const float *v,
                          float lr, float c1, float c2, float eps, size_t size);
/* cublasSaxpy   (src/cuda_utils.hpp:326-340)  B += alpha*A */
void mggcn_axpy_f32(mggcn_stream_t stream, const float *A, float *B, float alpha, size_t size);
/* cublasSscal   (src/cuda_utils.hpp:373-381)  mat *= scalar */
void mggcn_scale_mat_f32(mggcn_stream_t stream, float *mat, float scalar, size_t size);
/* cublasSasum   (src/cuda_utils.hpp:362-371)  *result_device = sum |A[i]|.
 * Unlike cuBLAS' host-pointer mode this does NOT block: the sum lands in device
 * memory on `stream` (fixed-order two-level reduction -> reproducible); the host
 * wrapper copies it back after its own sync (src/gcn.hpp:816-817). */
void mggcn_abssum_f32(mggcn_stream_t stream, const float *A, size_t size, float *result_device);

/* Halo pack: dst[k, 0:d] = src[indices[k], 0:d] for k < n_indices.
 * The reference has no counterpart -- its exchange broadcasts whole shards
 * (src/dist_matrix.hpp:458-467); the rows packed here are the ones its data-prep script counts
 * as communication volume (test/data/prep.py:237-244).  indices: device, uint32. */
void mggcn_gather_rows_f32(mggcn_stream_t stream, const float *src, size_t ld_src, const uint32_t *indices,
                           size_t n_indices, uint32_t d, float *dst, size_t ld_dst);

/* ======================================================================== *
 * Fused tail kernels.  Same math as the chains above,
 * fewer passes over [n x C] / fewer launches.
 * ======================================================================== */
/* softmax + argmax + log-prob + gradient in one pass over the logits, in place:
 * the 8-launch chain of softmax::operator() (src/gcn.hpp:651-675) +
 * softmax_cross_entropy_loss::operator() (src/gcn.hpp:785-818):
 *   O = softmax(H) row-wise (max-subtracted); P = argmax (first wins);
 *   loss_terms[i] = log O[i, Y[i]];  correct[i] = (Y[i] == P[i]);
 *   H <- (O - onehot(Y)) * grad_scale           (grad_scale = 1 / n_global)
 *   sums[0] += sum_i |loss_terms[i]|, sums[1] += sum_i correct[i]  (the caller zeroes sums;
 *   workgroup partials are added in a fixed order by a one-workgroup second launch:
 *   bitwise reproducible, no atomics). */
void mggcn_softmax_xent_fused_f32(mggcn_stream_t stream, float *H, const int32_t *Y, size_t n_rows,
                                  size_t m, float grad_scale, float *sums_device);
/* The same pass out of place: reads `logits`, writes the gradient to G (G == logits is the call above).  With
 * copy = true the reference copies the logits before its in-place chain (src/gcn.hpp:653-656, :787): here the copy
 * is the pass itself. */
void mggcn_softmax_xent_fused_from_f32(mggcn_stream_t stream, const float *logits, float *G, const int32_t *Y,
                                       size_t n_rows, size_t m, float grad_scale, float *sums_device);
/* One launch for linear::adam_update on a parameter tensor (src/gcn.hpp:146-172):
 *   g += wd*p; m = b1*m + (1-b1)*g; v = b2*v + (1-b2)*g*g;
 *   p -= (lr/c1) * m / (sqrt(v/c2) + eps)      (g is updated in place like the reference) */
void mggcn_adam_fused_f32(mggcn_stream_t stream, float *param, float *grad, float *m, float *v,
                          float lr, float beta1, float beta2, float weight_decay, float c1, float c2,
                          float eps, size_t size);

/* The same update for EVERY parameter tensor of a model in one launch (the reference runs seven launches per
 * layer, src/gcn.hpp:146-172: 28 per epoch on the 4-layer Reddit model; this is 1).  `table_device`: n_tensors
 * entries in DEVICE memory, built once (the pointers of a model never change); first_block = running sum of
 * mggcn_adam_multi_blocks(size) over the preceding entries, total_blocks = the sum over all.  lr / beta / c1 /
 * c2 / eps are shared (every layer takes the same step count); weight_decay is per tensor (0 for biases).
 * Element-wise math identical to mggcn_adam_fused_f32 -> bitwise equal results. */
typedef struct {
    float *param, *grad, *m, *v;
    uint64_t size;
    float weight_decay;
    uint32_t first_block;
} mggcn_adam_tensor;
uint32_t mggcn_adam_multi_blocks(uint64_t size);
void mggcn_adam_multi_f32(mggcn_stream_t stream, const mggcn_adam_tensor *table_device, uint32_t n_tensors,
                          uint32_t total_blocks, float lr, float beta1, float beta2, float c1, float c2, float eps);

/* ======================================================================== *
 * Host-side graph preprocessing (multi-threaded CPU code, runs once per dataset;
 * the reference does the same work on the host with parallel STL).
 * All pointers are HOST pointers.
 * ======================================================================== */
/* csr_matrix::normalize(axis) (src/matrix.hpp:340-390): axis==0 row-normalise,
 * axis!=0 column-normalise (what gcn uses, src/gcn.hpp:947).  Race-free and
 * deterministic (the reference's parallel column sums race, src/matrix.hpp:353-357). */
void mggcn_csr_normalize_host(uint32_t n_rows, uint32_t n_cols, const uint32_t *indptr,
                              const uint32_t *indices, float *values, int axis);
/* csr_matrix::transpose() (src/matrix.hpp:392-453).  Outputs: t_indptr[n_cols+1],
 * t_indices[nnz], t_values[nnz].  Within a transposed row, entries are in increasing
 * source-row order (the reference's serial order; its parallel path is unordered). */
void mggcn_csr_transpose_host(uint32_t n_rows, uint32_t n_cols, const uint32_t *indptr,
                              const uint32_t *indices, const float *values, uint32_t *t_indptr,
                              uint32_t *t_indices, float *t_values);
/* dist_row_csr_matrix ctor for one row block (src/dist_matrix.hpp:215-259):
 * rows [row_begin,row_end) are cut at the column boundaries q[0..nq] into nq CSR
 * blocks with block-local column indices.
 *   count: blk_indptr is nq*(rows+1) u32, block j at offset j*(rows+1)
 *   fill : blk_indices[j] / blk_values[j] sized blk_indptr[j][rows] */
void mggcn_csr_block_split_count_host(const uint32_t *indptr, const uint32_t *indices,
                                      uint32_t row_begin, uint32_t row_end, const uint32_t *q,
                                      uint32_t nq, uint32_t *blk_indptr);
void mggcn_csr_block_split_fill_host(const uint32_t *indptr, const uint32_t *indices,
                                     const float *values, uint32_t row_begin, uint32_t row_end,
                                     const uint32_t *q, uint32_t nq, const uint32_t *blk_indptr,
                                     uint32_t *const *blk_indices, float *const *blk_values);
/* dn_matrix::init(gain) (src/matrix.hpp:539-545): std::default_random_engine(99),
 * U(-g, g) with g = gain*sqrt(3/n_rows), row-major fill.  gain < 0 selects the
 * reference default sqrt(2/(1+0.01^2)). */
void mggcn_init_uniform_host(float *buffer, size_t n_rows, size_t n_cols, float gain);

#ifdef __cplusplus
}
#endif
#endif /* MGGCN_H_ */
