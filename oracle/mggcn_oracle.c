/*
 * mggcn_oracle.c -- CPU restatement of the MG-GCN hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the *checker* for the HIP path in mg-gcn_amd/csrc.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The
 * product never calls into it (the product aborts when the HIP library is
 * missing; there is no CPU fallback).
 *
 * Every function cites the reference lines (relative to /root/reference) whose
 * semantics it restates.  The reference's SpMM / GEMM arithmetic lives in
 * closed-source cuSPARSE / cuBLAS (src/cuda_utils.hpp:31, :169), so those two
 * are restated from their documented contract C = alpha*op(A)*op(B) + beta*C
 * (src/cuda_utils.hpp:15-26, :149-157) and pinned by the reference's own
 * known-answer tests (test/test_gcn.cpp:98-249) -- see tests/test_oracle_kat.py.
 *
 * Plain C11 + OpenMP; fp32 storage.  Accumulation is fp32 in the *_f32
 * functions (what the reference computes) and fp64 in the *_f64acc twins
 * (ground truth for error measurements).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_API __attribute__((visibility("default")))

ORC_API int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

ORC_API void orc_set_num_threads(int n) {
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* ------------------------------------------------------------------------- *
 * SpMM: C = alpha * A * B + beta * C, A in CSR (u32 indptr / u32 indices /
 * f32 values), B [A.m x d] and C [A.n x d] row-major with leading dimensions
 * ldb / ldc.  Reference: src/cuda_utils.hpp:15-32 (cusparseSpMM, op N/N,
 * CUSPARSE_ORDER_ROW src/matrix.hpp:508, 32-bit indices src/matrix.hpp:271).
 * Row-parallel in the idiom of src/matrix.hpp:342-349; within a row the
 * non-zeros are accumulated in storage order.
 * beta == 0 never reads C (cuSPARSE contract: C may be uninitialised).
 * ------------------------------------------------------------------------- */
ORC_API void orc_spmm_csr_f32(uint32_t n_rows, const uint32_t *indptr, const uint32_t *indices,
                              const float *vals, const float *B, size_t ldb, float *C, size_t ldc,
                              uint32_t d, float alpha, float beta) {
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t r = 0; r < (int64_t)n_rows; r++) {
        float acc[1024];
        float *c = C + (size_t)r * ldc;
        for (uint32_t j0 = 0; j0 < d; j0 += 1024) {
            const uint32_t w = d - j0 < 1024 ? d - j0 : 1024;
            for (uint32_t j = 0; j < w; j++) acc[j] = 0.f;
            for (uint32_t e = indptr[r]; e < indptr[r + 1]; e++) {
                const float v = vals[e];
                const float *b = B + (size_t)indices[e] * ldb + j0;
#pragma omp simd
                for (uint32_t j = 0; j < w; j++) acc[j] += v * b[j];
            }
            if (beta == 0.f)
                for (uint32_t j = 0; j < w; j++) c[j0 + j] = alpha * acc[j];
            else
                for (uint32_t j = 0; j < w; j++) c[j0 + j] = alpha * acc[j] + beta * c[j0 + j];
        }
    }
}

/* fp64-accumulating twin: the error yardstick (|gpu - f64| vs |f32oracle - f64|). */
ORC_API void orc_spmm_csr_f64acc(uint32_t n_rows, const uint32_t *indptr, const uint32_t *indices,
                                 const float *vals, const float *B, size_t ldb, float *C, size_t ldc,
                                 uint32_t d, float alpha, float beta) {
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t r = 0; r < (int64_t)n_rows; r++) {
        double acc[1024];
        float *c = C + (size_t)r * ldc;
        for (uint32_t j0 = 0; j0 < d; j0 += 1024) {
            const uint32_t w = d - j0 < 1024 ? d - j0 : 1024;
            for (uint32_t j = 0; j < w; j++) acc[j] = 0.0;
            for (uint32_t e = indptr[r]; e < indptr[r + 1]; e++) {
                const double v = vals[e];
                const float *b = B + (size_t)indices[e] * ldb + j0;
                for (uint32_t j = 0; j < w; j++) acc[j] += v * (double)b[j];
            }
            for (uint32_t j = 0; j < w; j++) {
                double o = (double)alpha * acc[j];
                if (beta != 0.f) o += (double)beta * (double)c[j0 + j];
                c[j0 + j] = (float)o;
            }
        }
    }
}

/* ------------------------------------------------------------------------- *
 * csr_matrix::normalize(axis) -- src/matrix.hpp:340-390.
 * axis == 0: every value divided by its row sum; axis != 0: by its column sum
 * (D_col[c] = sum_r A[r,c]).  The reference's parallel column accumulation is
 * racy (src/matrix.hpp:353-357); restated race-free in the order of the serial
 * body (src/matrix.hpp:378-388): row-major traversal, fp32 accumulation.
 * ------------------------------------------------------------------------- */
ORC_API void orc_csr_normalize(uint32_t n, uint32_t m, const uint32_t *indptr,
                               const uint32_t *indices, float *data, int axis) {
    if (!axis) {
        for (uint32_t v = 0; v < n; v++) {
            float sum = 0.f;
            for (uint32_t e = indptr[v]; e < indptr[v + 1]; e++) sum += data[e];
            for (uint32_t e = indptr[v]; e < indptr[v + 1]; e++) data[e] /= sum;
        }
    } else {
        float *deg = (float *)calloc(m ? m : 1, sizeof(float));
        for (uint32_t v = 0; v < n; v++)
            for (uint32_t e = indptr[v]; e < indptr[v + 1]; e++) deg[indices[e]] += data[e];
        for (uint32_t v = 0; v < n; v++)
            for (uint32_t e = indptr[v]; e < indptr[v + 1]; e++) data[e] /= deg[indices[e]];
        free(deg);
    }
}

/* ------------------------------------------------------------------------- *
 * csr_matrix::transpose() -- src/matrix.hpp:392-453 (counting sort).  Restated
 * from the serial body (:426-452): within a transposed row entries appear in
 * increasing source-row order (the parallel version's order is
 * nondeterministic, src/matrix.hpp:404-408).
 * t_indptr has m+1 entries, t_indices/t_data nnz entries.
 * ------------------------------------------------------------------------- */
ORC_API void orc_csr_transpose(uint32_t n, uint32_t m, const uint32_t *indptr,
                               const uint32_t *indices, const float *data, uint32_t *t_indptr,
                               uint32_t *t_indices, float *t_data) {
    const uint32_t nnz = indptr[n] - indptr[0];
    uint32_t *dloc = (uint32_t *)malloc((nnz ? nnz : 1) * sizeof(uint32_t));
    memset(t_indptr, 0, ((size_t)m + 1) * sizeof(uint32_t));
    for (uint32_t i = 0; i < n; i++)
        for (uint32_t j = indptr[i]; j < indptr[i + 1]; j++) dloc[j] = t_indptr[indices[j] + 1]++;
    for (uint32_t c = 0; c < m; c++) t_indptr[c + 1] += t_indptr[c];
    for (uint32_t i = 0; i < n; i++)
        for (uint32_t j = indptr[i]; j < indptr[i + 1]; j++) {
            const uint32_t loc = t_indptr[indices[j]] + dloc[j];
            t_indices[loc] = i;
            t_data[loc] = data[j];
        }
    free(dloc);
}

/* csr_matrix::as_dn() -- src/matrix.hpp:328-337 (duplicates: last one wins). */
ORC_API void orc_csr_as_dn(uint32_t n, uint32_t m, const uint32_t *indptr, const uint32_t *indices,
                           const float *data, float *out) {
    memset(out, 0, (size_t)n * m * sizeof(float));
    for (uint32_t v = 0; v < n; v++)
        for (uint32_t e = indptr[v]; e < indptr[v + 1]; e++)
            out[(size_t)v * m + indices[e]] = data[e];
}

/* ------------------------------------------------------------------------- *
 * dist_row_csr_matrix ctor -- src/dist_matrix.hpp:215-259 (+ lowerbound_index
 * :177-185).  Splits rows [p[i], p[i+1]) into one CSR block per column range
 * [q[j], q[j+1]); column indices re-based to q[j]; within-row order preserved.
 *
 * Two-call protocol for one row block i:
 *   orc_block_split_count : fills blk_indptr[j][0..rows] (each rows+1 long,
 *                           concatenated: nq blocks) -> caller sizes outputs
 *   orc_block_split_fill  : fills blk_indices / blk_data at the given offsets
 * ------------------------------------------------------------------------- */
static uint32_t orc_col_block(const uint32_t *q, uint32_t nq, uint32_t col) {
    /* index j with q[j] <= col < q[j+1]  (std::lower_bound(q, col+1) - 1) */
    uint32_t lo = 0, hi = nq; /* q has nq+1 entries */
    while (lo < hi) {
        uint32_t mid = (lo + hi + 1) / 2;
        if (q[mid] <= col) lo = mid; else hi = mid - 1;
    }
    return lo;
}

ORC_API void orc_block_split_count(const uint32_t *indptr, const uint32_t *indices, uint32_t row_beg,
                                   uint32_t row_end, const uint32_t *q, uint32_t nq,
                                   uint32_t *blk_indptr /* nq * (rows+1) */) {
    const uint32_t rows = row_end - row_beg;
    memset(blk_indptr, 0, (size_t)nq * (rows + 1) * sizeof(uint32_t));
    for (uint32_t r = 0; r < rows; r++)
        for (uint32_t k = indptr[row_beg + r]; k < indptr[row_beg + r + 1]; k++)
            blk_indptr[(size_t)orc_col_block(q, nq, indices[k]) * (rows + 1) + r + 1]++;
    for (uint32_t j = 0; j < nq; j++) {
        uint32_t *p = blk_indptr + (size_t)j * (rows + 1);
        for (uint32_t r = 0; r < rows; r++) p[r + 1] += p[r];
    }
}

ORC_API void orc_block_split_fill(const uint32_t *indptr, const uint32_t *indices, const float *data,
                                  uint32_t row_beg, uint32_t row_end, const uint32_t *q, uint32_t nq,
                                  const uint32_t *blk_indptr, uint32_t *const *blk_indices,
                                  float *const *blk_data) {
    const uint32_t rows = row_end - row_beg;
    uint32_t *fill = (uint32_t *)calloc((size_t)nq, sizeof(uint32_t));
    for (uint32_t r = 0; r < rows; r++) {
        memset(fill, 0, (size_t)nq * sizeof(uint32_t));
        for (uint32_t k = indptr[row_beg + r]; k < indptr[row_beg + r + 1]; k++) {
            const uint32_t j = orc_col_block(q, nq, indices[k]);
            const uint32_t at = blk_indptr[(size_t)j * (rows + 1) + r] + fill[j]++;
            blk_indices[j][at] = indices[k] - q[j];
            blk_data[j][at] = data[k];
        }
    }
    free(fill);
}

/* ------------------------------------------------------------------------- *
 * Dense GEMM, row-major: C = alpha * op(A) * op(B) + beta * C.
 * Reference: src/cuda_utils.hpp:149-172 (cublasSgemm with swapped operands to
 * emulate row-major).  M,N,K are the op()-ed shapes: op(A) is MxK, op(B) KxN.
 * lda/ldb/ldc are the stored leading dimensions.
 * ------------------------------------------------------------------------- */
ORC_API void orc_gemm_f32(int trans_a, int trans_b, uint32_t M, uint32_t N, uint32_t K, float alpha,
                          const float *A, size_t lda, const float *B, size_t ldb, float beta,
                          float *C, size_t ldc) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)M; i++) {
        float *c = C + (size_t)i * ldc;
        float *acc = (float *)malloc((size_t)N * sizeof(float));
        for (uint32_t j = 0; j < N; j++) acc[j] = 0.f;
        for (uint32_t k = 0; k < K; k++) {
            const float a = trans_a ? A[(size_t)k * lda + i] : A[(size_t)i * lda + k];
            if (!trans_b) {
                const float *b = B + (size_t)k * ldb;
#pragma omp simd
                for (uint32_t j = 0; j < N; j++) acc[j] += a * b[j];
            } else {
                for (uint32_t j = 0; j < N; j++) acc[j] += a * B[(size_t)j * ldb + k];
            }
        }
        for (uint32_t j = 0; j < N; j++)
            c[j] = beta == 0.f ? alpha * acc[j] : alpha * acc[j] + beta * c[j];
        free(acc);
    }
}

ORC_API void orc_gemm_f64acc(int trans_a, int trans_b, uint32_t M, uint32_t N, uint32_t K,
                             float alpha, const float *A, size_t lda, const float *B, size_t ldb,
                             float beta, float *C, size_t ldc) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)M; i++) {
        float *c = C + (size_t)i * ldc;
        double *acc = (double *)calloc((size_t)N, sizeof(double));
        for (uint32_t k = 0; k < K; k++) {
            const double a = trans_a ? A[(size_t)k * lda + i] : A[(size_t)i * lda + k];
            for (uint32_t j = 0; j < N; j++)
                acc[j] += a * (double)(trans_b ? B[(size_t)j * ldb + k] : B[(size_t)k * ldb + j]);
        }
        for (uint32_t j = 0; j < N; j++) {
            double o = (double)alpha * acc[j];
            if (beta != 0.f) o += (double)beta * (double)c[j];
            c[j] = (float)o;
        }
        free(acc);
    }
}

/* ------------------------------------------------------------------------- *
 * Element-wise / row kernels -- restated one for one from the live kernels of
 * src/cuda_utils.cu (line numbers per function).  "size" is the element count
 * n*m as in the reference launchers (src/cuda_utils.cu:229-390).
 * ------------------------------------------------------------------------- */

/* src/cuda_utils.cu:26-31  out = max(in, alpha*in) */
ORC_API void orc_leaky_relu_forward(const float *in, float *out, size_t size, float alpha) {
    for (size_t i = 0; i < size; i++) {
        const float x = in[i], y = alpha * x;
        out[i] = x > y ? x : y;
    }
}

/* src/cuda_utils.cu:33-38  G_out = in > 0 ? G_in : alpha*G_in  (in = activated output) */
ORC_API void orc_leaky_relu_backward(const float *in, const float *G_in, float *G_out, size_t size,
                                     float alpha) {
    for (size_t i = 0; i < size; i++) G_out[i] = in[i] > 0.f ? G_in[i] : alpha * G_in[i];
}

/* src/cuda_utils.cu:40-51  mat[i,:] (=|+=) row */
ORC_API void orc_broadcast_rows(const float *row, float *mat, size_t size, size_t m, int discard) {
    for (size_t i = 0; i < size; i++) {
        if (discard) mat[i] = row[i % m]; else mat[i] += row[i % m];
    }
}

/* src/cuda_utils.cu:75-79  mat[i] /= scalar[i / m] */
ORC_API void orc_scale_rows(float *mat, const float *scalar, size_t size, size_t m) {
    for (size_t i = 0; i < size; i++) mat[i] /= scalar[i / m];
}

/* src/cuda_utils.cu:95-104  row max */
ORC_API void orc_max_rows(const float *mat, float *maxs, size_t size, size_t m) {
    for (size_t row = 0; row * m < size; row++) {
        float mx = -INFINITY;
        for (size_t i = 0; i < m; i++) mx = mx > mat[row * m + i] ? mx : mat[row * m + i];
        maxs[row] = mx;
    }
}

/* src/cuda_utils.cu:119-133  argmax, first maximum wins (strict <) */
ORC_API void orc_max_row_indices(const float *mat, int32_t *maxs, size_t size, size_t m) {
    for (size_t row = 0; row * m < size; row++) {
        float mx = -INFINITY;
        size_t index = 0;
        for (size_t i = 0; i < m; i++)
            if (mx < mat[row * m + i]) { mx = mat[row * m + i]; index = i; }
        maxs[row] = (int32_t)index;
    }
}

/* src/cuda_utils.cu:142-150  values[row] = log(mat[row, indices[row]]) */
ORC_API void orc_index_log_rows(const float *mat, const int32_t *indices, float *values, size_t size,
                                size_t m) {
    for (size_t row = 0; row * m < size; row++) values[row] = logf(mat[row * m + indices[row]]);
}

/* src/cuda_utils.cu:159-164  mat[row, indices[row]] += alpha */
ORC_API void orc_add_indexed_rows(float *mat, const int32_t *indices, float alpha, size_t size,
                                  size_t m) {
    for (size_t row = 0; row * m < size; row++) mat[row * m + indices[row]] += alpha;
}

/* src/cuda_utils.cu:180-184  out[i] = (mat1[i] == mat2[i]) */
ORC_API void orc_is_equal(const int32_t *mat1, const int32_t *mat2, float *out, size_t size) {
    for (size_t i = 0; i < size; i++) out[i] = (float)(mat1[i] == mat2[i]);
}

/* src/cuda_utils.cu:192-200  out = exp(mat - scalar[row]) */
ORC_API void orc_subtract_rows_exp(const float *mat, const float *scalar, float *out, size_t size,
                                   size_t m) {
    for (size_t i = 0; i < size; i++) out[i] = expf(mat[i] - scalar[i / m]);
}

/* src/cuda_utils.cu:81-86  B = alpha*A + beta*B */
ORC_API void orc_axpby(const float *A, float *B, float alpha, float beta, size_t size) {
    for (size_t i = 0; i < size; i++) B[i] = alpha * A[i] + beta * B[i];
}

/* src/cuda_utils.cu:88-93  B = alpha*A*A + beta*B */
ORC_API void orc_aaxpby(const float *A, float *B, float alpha, float beta, size_t size) {
    for (size_t i = 0; i < size; i++) B[i] = alpha * A[i] * A[i] + beta * B[i];
}

/* src/cuda_utils.cu:208-218 with the launcher's lr/c1 (src/cuda_utils.cu:387):
 * param -= (lr/c1) * m / (sqrt(v/c2) + eps) */
ORC_API void orc_adam_final(float *param, const float *m, const float *v, float lr, float c1,
                            float c2, float eps, size_t size) {
    const float step = lr / c1;
    for (size_t i = 0; i < size; i++) param[i] -= step * m[i] / (sqrtf(v[i] / c2) + eps);
}

/* cublasSaxpy src/cuda_utils.cu -> src/cuda_utils.hpp:326-340  B += alpha*A */
ORC_API void orc_axpy(const float *A, float *B, float alpha, size_t size) {
    for (size_t i = 0; i < size; i++) B[i] += alpha * A[i];
}

/* cublasSscal src/cuda_utils.hpp:373-381 */
ORC_API void orc_scale_mat(float *mat, float scalar, size_t size) {
    for (size_t i = 0; i < size; i++) mat[i] *= scalar;
}

/* cublasSasum src/cuda_utils.hpp:362-371; fp64 accumulation for a stable yardstick */
ORC_API float orc_abssum(const float *A, size_t size) {
    double s = 0.0;
    for (size_t i = 0; i < size; i++) s += fabs((double)A[i]);
    return (float)s;
}
