// ops.hpp -- the reference's operator overload set (src/cuda_utils.hpp) over the C ABI.
//
// Same free-function names and argument order: matmul / get_matmul_buffer (SpMM :27-32,
// :94-102; GEMM :158-172; distributed SpMM :47-92; distributed row GEMMs :304-324), axpy,
// scale_mat, abssum, and the kernel wrappers (:470-748).  Each wrapper enqueues one C-ABI
// call on the context's compute stream.  Shape preconditions (assert in the reference)
// throw std::invalid_argument.
#pragma once

#include <memory>
#include <string>
#include <vector>

#include "dist_matrix.hpp"
#include "matrix.hpp"

inline void mggcn_require(bool ok, const char *what) {
    if (!ok) throw std::invalid_argument(what);
}

// the reference's opaque cuSPARSE workspace (cuda_ptr<char>) becomes the SpMM plan
using spmm_buffer = std::shared_ptr<mggcn_spmm_plan>;

// ---- SpMM ------------------------------------------------------------------------------
template <typename x_t, typename v_t, typename r_t>
spmm_buffer get_matmul_buffer(const context ctx, const csr_matrix<x_t, v_t, r_t> A, const dn_matrix<r_t> B,
                              const dn_matrix<r_t> C, const r_t = 1, const r_t = 0) {
    mggcn_require(A.m() == B.n() && A.n() == C.n() && B.m() == C.m(), "get_matmul_buffer: shape mismatch");
    ctx.set();
    return spmm_buffer(mggcn_spmm_plan_create_for(A.n(), A.m(), A.indptr().data(), A.indices().data(), A.data().data(),
                                                  (uint32_t)std::max<std::size_t>(B.m(), 128), (uint32_t)B.m()),
                       &mggcn_spmm_plan_destroy);
}

template <typename x_t, typename v_t, typename r_t>
void matmul(const context ctx, const csr_matrix<x_t, v_t, r_t> A, const dn_matrix<r_t> B, const dn_matrix<r_t> C,
            const spmm_buffer ext_buffer, const r_t alpha, const r_t beta, const uint32_t flags = MGGCN_SPMM_DEFAULT,
            const r_t slope = 0.01f) {
    mggcn_require(A.m() == B.n() && B.m() == C.m() && A.n() == C.n(), "matmul(csr): shape mismatch");
    ctx.set();
    auto [indptr, indices, data] = A.buffer();
    mggcn_spmm_csr_f32(ctx.stream(0), ext_buffer.get(), A.n(), A.m(), indptr.get(), indices.get(), data.get(),
                       B.buffer(), B.m(), C.buffer(), C.m(), (uint32_t)B.m(), alpha, beta, flags, slope);
}

// ---- GEMM ------------------------------------------------------------------------------
template <typename r_t>
void matmul(const context ctx, const dn_matrix<r_t> A, const dn_matrix<r_t> B, const dn_matrix<r_t> C, const r_t alpha,
            const r_t beta, const bool A_T = false, const bool B_T = false) {
    auto A_n = A.n(), A_m = A.m(), B_n = B.n(), B_m = B.m();
    if (A_T) std::swap(A_n, A_m);
    if (B_T) std::swap(B_n, B_m);
    mggcn_require(A_m == B_n && A_n == C.n() && B_m == C.m(), "matmul(dn): shape mismatch");
    ctx.set();
    const auto ws = mggcn_gemm_workspace_bytes(A_T, B_T, (uint32_t)A_n, (uint32_t)B_m, (uint32_t)A_m);
    mggcn_gemm_f32(ctx.stream(0), A_T, B_T, (uint32_t)A_n, (uint32_t)B_m, (uint32_t)A_m, alpha, A.buffer(), A.m(),
                   B.buffer(), B.m(), beta, C.buffer(), C.m(), ctx.gemm_workspace(ws), ws);
}

// XW = X.W + 1 b^T with the bias in the GEMM epilogue (fused form of src/gcn.hpp:116-123)
template <typename r_t>
void linear_forward(const context ctx, const dn_matrix<r_t> X, const dn_matrix<r_t> W, const dn_matrix<r_t> b,
                    const dn_matrix<r_t> XW) {
    mggcn_require(X.m() == W.n() && XW.n() == X.n() && XW.m() == W.m() && b.m() == W.m() && b.n() == 1,
                  "linear_forward: shape mismatch");
    ctx.set();
    const auto ws = mggcn_gemm_workspace_bytes(0, 0, (uint32_t)X.n(), (uint32_t)W.m(), (uint32_t)X.m());
    mggcn_gemm_bias_f32(ctx.stream(0), 0, 0, (uint32_t)X.n(), (uint32_t)W.m(), (uint32_t)X.m(), (r_t)1, X.buffer(), X.m(),
                        W.buffer(), W.m(), b.buffer(), XW.buffer(), XW.m(), ctx.gemm_workspace(ws), ws);
}

// ---- distributed SpMM: C_j = beta C_j + alpha sum_i A[j,i] B_i ------------------------------
struct dist_spmm_buffers {
    std::vector<std::vector<spmm_buffer>> block;   // [j][i]  (rounds schedule)
    std::vector<spmm_buffer> remote;               // [j]     (all-gather schedule; diagonal = block[j][j])
};

template <typename x_t, typename v_t, typename r_t>
dist_spmm_buffers get_matmul_buffer(const dist_context ctx, const dist_row_csr_matrix<x_t, v_t, r_t> A,
                                    const dist_row_dn_matrix<r_t> B, const dist_row_dn_matrix<r_t> C, bool rounds) {
    dist_spmm_buffers out;
    const auto P = ctx.size();
    out.block.resize(P);
    for (std::size_t j = 0; j < P; j++) {
        ctx[j].set();
        out.block[j].resize(P);
        for (std::size_t i = 0; i < P; i++)
            if (rounds || i == j) {
                const auto blk = A[{j, i}];
                out.block[j][i] = spmm_buffer(mggcn_spmm_plan_create_for(blk.n(), blk.m(), blk.indptr().data(), blk.indices().data(),
                                                                         blk.data().data(), (uint32_t)std::max<std::size_t>(B.m(), 128),
                                                                         (uint32_t)B.m()),
                                              &mggcn_spmm_plan_destroy);
            }
        if (!rounds) {
            const auto &rem = A.remote(j);
            out.remote.push_back(spmm_buffer(mggcn_spmm_plan_create_for(rem.n(), rem.m(), rem.indptr().data(), rem.indices().data(),
                                                                        rem.data().data(), (uint32_t)std::max<std::size_t>(B.m(), 128),
                                                                        (uint32_t)B.m()),
                                             &mggcn_spmm_plan_destroy));
        }
    }
    (void)C;
    return out;
}

// The reference's pipelined schedule (src/cuda_utils.hpp:57-92): round i broadcasts shard i
// on the comm stream while the compute stream multiplies block column i-1; two receive
// buffers; events order the hand-offs; per-round timers.
template <typename x_t, typename v_t, typename r_t>
void matmul(dist_context ctx, dist_row_csr_matrix<x_t, v_t, r_t> A, dist_row_dn_matrix<r_t> B, dist_row_dn_matrix<r_t> C,
            const dist_spmm_buffers &ext, std::vector<dist_row_dn_matrix<r_t>> B_bcast, const r_t alpha, const r_t beta,
            const std::string name = "", const uint32_t last_flags = 0) {
    const auto P = ctx.size();
    ctx.record(name + "0_matmul-spmm", 0);
    ctx.wait(name + "0_matmul-spmm", 1);
    for (std::size_t i = 0; i < P; i++) {
        if (i > 1) ctx.wait(name + std::to_string(i - 1) + "_matmul-spmm", ctx.bcast_stream_id());   // double-buffer hazard
        ctx.record(name + std::to_string(i) + "_matmul-bcast-start", ctx.bcast_stream_id());
        B.bcast(ctx, i, B_bcast[i % 2], ctx.bcast_stream_id());
        ctx.record(name + std::to_string(i) + "_matmul-bcast-finish", ctx.bcast_stream_id());
        ctx.wait(name + std::to_string(i) + "_matmul-bcast-finish", 0);
        for (std::size_t j = 0; j < P; j++)
            matmul(ctx[j], A[{j, i}], B_bcast[i % 2][j], C[j], ext.block[j][i], alpha, i == 0 ? beta : (r_t)1,
                   i + 1 == P ? last_flags : 0u);
        ctx.record(name + std::to_string(i + 1) + "_matmul-spmm", 0);
    }
    ctx.register_timer(name + "matmul-spmm", name + "0_matmul-spmm", name + std::to_string(P) + "_matmul-spmm");
}

// MI355X-first schedule: ONE all-gather of the shards on the comm stream, overlapped with
// the SpMM of the diagonal block (no dependency); the merged remote blocks follow (beta = 1).
template <typename x_t, typename v_t, typename r_t>
void matmul_allgather(dist_context ctx, dist_row_csr_matrix<x_t, v_t, r_t> A, dist_row_dn_matrix<r_t> B,
                      dist_row_dn_matrix<r_t> C, const dist_spmm_buffers &ext, const std::vector<dn_matrix<r_t>> &gathered,
                      const r_t alpha, const r_t beta, const std::string name = "", const uint32_t last_flags = 0) {
    const auto P = ctx.size();
    const int cs = ctx.bcast_stream_id();
    ctx.record(name + "0_matmul-spmm", 0);
    ctx.wait(name + "0_matmul-spmm", 1);
    ctx.record(name + "0_matmul-bcast-start", cs);
    if (P > 1) B.allgather(ctx, gathered, cs);
    ctx.record(name + "0_matmul-bcast-finish", cs);
    for (std::size_t j = 0; j < P; j++)
        matmul(ctx[j], A[{j, j}], B[j], C[j], ext.block[j][j], alpha, beta, P == 1 ? last_flags : 0u);
    if (P > 1) {
        ctx.wait(name + "0_matmul-bcast-finish", 0);
        for (std::size_t j = 0; j < P; j++)
            matmul(ctx[j], A.remote(j), gathered[j], C[j], ext.remote[j], alpha, (r_t)1, last_flags);
    }
    ctx.record(name + "1_matmul-spmm", 0);
    ctx.register_timer(name + "matmul-spmm", name + "0_matmul-spmm", name + "1_matmul-spmm");
}

// ---- distributed row GEMMs (reference src/cuda_utils.hpp:304-324) ---------------------------
template <typename r_t>
void matmul(const dist_context ctx, const dist_row_dn_matrix<r_t> A, const dist_row_dn_matrix<r_t> B,
            const repl_dn_matrix<r_t> C, const r_t alpha, const r_t beta) {      // C = sum_i A_i^T B_i
    for (std::size_t i = 0; i < ctx.size(); i++) matmul(ctx[i], A[i], B[i], C[i], alpha, beta, true);
    C.allreduce(ctx);
}

template <typename r_t>
void matmul(const dist_context ctx, const dist_row_dn_matrix<r_t> A, const repl_dn_matrix<r_t> B,
            const dist_row_dn_matrix<r_t> C, const r_t alpha, const r_t beta, const bool B_T = false) {
    for (std::size_t i = 0; i < ctx.size(); i++) matmul(ctx[i], A[i], B[i], C[i], alpha, beta, false, B_T);
}

template <typename r_t>
void linear_forward(const dist_context ctx, const dist_row_dn_matrix<r_t> X, const repl_dn_matrix<r_t> W,
                    const repl_dn_matrix<r_t> b, const dist_row_dn_matrix<r_t> XW) {
    for (std::size_t i = 0; i < ctx.size(); i++) linear_forward(ctx[i], X[i], W[i], b[i], XW[i]);
}

// ---- BLAS-1 (reference src/cuda_utils.hpp:326-381) -------------------------------------------
template <typename r_t>
void axpy(const context ctx, const dn_matrix<r_t> A, const dn_matrix<r_t> B, const r_t alpha) {
    mggcn_require(A.shape() == B.shape(), "axpy: shape mismatch");
    ctx.set();
    mggcn_axpy_f32(ctx.stream(0), A.buffer(), B.buffer(), alpha, A.size());
}
template <typename r_t>
void scale_mat(const context ctx, const dn_matrix<r_t> mat, r_t scalar) {
    ctx.set();
    mggcn_scale_mat_f32(ctx.stream(0), mat.buffer(), scalar, mat.size());
}
// *result_device = sum |A| (enqueue-only; cublasSasum's host-pointer mode blocks instead)
template <typename r_t>
void abssum(const context ctx, const dn_matrix<r_t> A, r_t *result_device) {
    ctx.set();
    mggcn_abssum_f32(ctx.stream(0), A.buffer(), A.size(), result_device);
}

// ---- kernel wrappers (reference src/cuda_utils.hpp:470-748) ----------------------------------
template <typename r_t>
void leaky_relu_forward(const context ctx, const dn_matrix<r_t> in, const dn_matrix<r_t> out, r_t alpha = 0.01) {
    mggcn_require(in.shape() == out.shape(), "leaky_relu_forward: shape mismatch");
    ctx.set();
    mggcn_leaky_relu_forward_f32(ctx.stream(0), in.buffer(), out.buffer(), in.size(), alpha);
}
template <typename r_t>
void leaky_relu_backward(const context ctx, const dn_matrix<r_t> in, const dn_matrix<r_t> G_in, const dn_matrix<r_t> G_out,
                         r_t alpha = 0.01) {
    mggcn_require(in.shape() == G_in.shape() && in.shape() == G_out.shape(), "leaky_relu_backward: shape mismatch");
    ctx.set();
    mggcn_leaky_relu_backward_f32(ctx.stream(0), in.buffer(), G_in.buffer(), G_out.buffer(), in.size(), alpha);
}
template <typename r_t>
void broadcast_rows(const context ctx, const dn_matrix<r_t> row, const dn_matrix<r_t> mat, const bool discard = true) {
    mggcn_require(row.m() == mat.m(), "broadcast_rows: width mismatch");
    ctx.set();
    mggcn_broadcast_rows_f32(ctx.stream(0), row.buffer(), mat.buffer(), mat.size(), mat.m(), discard);
}
template <typename r_t>
void scale_rows(const context ctx, const dn_matrix<r_t> mat, const dn_matrix<r_t> scalar) {
    mggcn_require(mat.n() == scalar.n(), "scale_rows: row count mismatch");
    ctx.set();
    mggcn_scale_rows_f32(ctx.stream(0), mat.buffer(), scalar.buffer(), mat.size(), mat.m());
}
template <typename r_t>
void max_rows(const context ctx, const dn_matrix<r_t> mat, const dn_matrix<r_t> maxs) {
    mggcn_require(mat.n() == maxs.n() && maxs.m() == 1, "max_rows: maxs must be n x 1");
    ctx.set();
    mggcn_max_rows_f32(ctx.stream(0), mat.buffer(), maxs.buffer(), mat.size(), mat.m());
}
template <typename r_t, typename x_t>
void max_row_indices(const context ctx, const dn_matrix<r_t> mat, const dn_matrix<x_t> maxs) {
    mggcn_require(mat.n() == maxs.n() && maxs.m() == 1, "max_row_indices: maxs must be n x 1");
    ctx.set();
    mggcn_max_row_indices_f32(ctx.stream(0), mat.buffer(), maxs.buffer(), mat.size(), mat.m());
}
template <typename r_t, typename x_t>
void index_log_rows(const context ctx, const dn_matrix<r_t> mat, const dn_matrix<x_t> indices, const dn_matrix<r_t> values) {
    mggcn_require(mat.n() == indices.n() && indices.m() == 1 && values.n() == mat.n() && values.m() == 1, "index_log_rows: shape");
    ctx.set();
    mggcn_index_log_rows_f32(ctx.stream(0), mat.buffer(), indices.buffer(), values.buffer(), mat.size(), mat.m());
}
template <typename r_t, typename x_t>
void add_indexed_rows(const context ctx, const dn_matrix<r_t> mat, const dn_matrix<x_t> indices, const r_t alpha) {
    mggcn_require(mat.n() == indices.n() && indices.m() == 1, "add_indexed_rows: shape");
    ctx.set();
    mggcn_add_indexed_rows_f32(ctx.stream(0), mat.buffer(), indices.buffer(), alpha, mat.size(), mat.m());
}
template <typename r_t, typename x_t>
void is_equal(const context ctx, const dn_matrix<x_t> a, const dn_matrix<x_t> b, const dn_matrix<r_t> out) {
    mggcn_require(a.shape() == b.shape() && a.shape() == out.shape(), "is_equal: shape mismatch");
    ctx.set();
    mggcn_is_equal_i32(ctx.stream(0), a.buffer(), b.buffer(), out.buffer(), a.size());
}
template <typename r_t>
void subtract_rows_exp(const context ctx, const dn_matrix<r_t> mat, const dn_matrix<r_t> scalar, const dn_matrix<r_t> out) {
    mggcn_require(mat.n() == scalar.n() && scalar.m() == 1 && mat.shape() == out.shape(), "subtract_rows_exp: shape");
    ctx.set();
    mggcn_subtract_rows_exp_f32(ctx.stream(0), mat.buffer(), scalar.buffer(), out.buffer(), mat.size(), mat.m());
}
template <typename r_t>
void axpby(const context ctx, const dn_matrix<r_t> A, const dn_matrix<r_t> B, const r_t alpha, const r_t beta) {
    mggcn_require(A.shape() == B.shape(), "axpby: shape mismatch");
    ctx.set();
    mggcn_axpby_f32(ctx.stream(0), A.buffer(), B.buffer(), alpha, beta, A.size());
}
template <typename r_t>
void aaxpby(const context ctx, const dn_matrix<r_t> A, const dn_matrix<r_t> B, const r_t alpha, const r_t beta) {
    mggcn_require(A.shape() == B.shape(), "aaxpby: shape mismatch");
    ctx.set();
    mggcn_aaxpby_f32(ctx.stream(0), A.buffer(), B.buffer(), alpha, beta, A.size());
}
template <typename r_t>
void adam_final(const context ctx, const dn_matrix<r_t> param, const dn_matrix<r_t> m, const dn_matrix<r_t> v, const r_t lr,
                const r_t c1, const r_t c2, const r_t eps) {
    mggcn_require(param.shape() == m.shape() && v.shape() == m.shape(), "adam_final: shape mismatch");
    ctx.set();
    mggcn_adam_final_f32(ctx.stream(0), param.buffer(), m.buffer(), v.buffer(), lr, c1, c2, eps, param.size());
}
// fused tail kernels (include/mggcn.h, "Fused tail kernels")
template <typename r_t>
void adam_fused(const context ctx, const dn_matrix<r_t> p, const dn_matrix<r_t> g, const dn_matrix<r_t> m, const dn_matrix<r_t> v,
                r_t lr, r_t b1, r_t b2, r_t wd, r_t c1, r_t c2, r_t eps) {
    ctx.set();
    mggcn_adam_fused_f32(ctx.stream(0), p.buffer(), g.buffer(), m.buffer(), v.buffer(), lr, b1, b2, wd, c1, c2, eps, p.size());
}
template <typename r_t, typename x_t>
void softmax_xent_fused(const context ctx, const dn_matrix<r_t> H, const dn_matrix<x_t> Y, r_t grad_scale, r_t *sums_device) {
    mggcn_require(H.n() == Y.n() && Y.m() == 1, "softmax_xent_fused: labels must be n x 1");
    ctx.set();
    mggcn_softmax_xent_fused_f32(ctx.stream(0), H.buffer(), Y.buffer(), H.n(), H.m(), grad_scale, sums_device);
}

// dist_context forms: per-GPU loops, as in the reference's "template<dn_t>" overloads
#define MGGCN_DIST_LOOP(call) for (std::size_t i = 0; i < ctx.size(); i++) { call; }
template <typename r_t, template <typename> class dn_t>
void leaky_relu_forward(const dist_context ctx, const dn_t<r_t> in, const dn_t<r_t> out, r_t a = 0.01) { MGGCN_DIST_LOOP(leaky_relu_forward(ctx[i], in[i], out[i], a)) }
template <typename r_t, template <typename> class dn_t>
void leaky_relu_backward(const dist_context ctx, const dn_t<r_t> in, const dn_t<r_t> G_in, const dn_t<r_t> G_out, r_t a = 0.01) { MGGCN_DIST_LOOP(leaky_relu_backward(ctx[i], in[i], G_in[i], G_out[i], a)) }
template <typename r_t, template <typename> class d1, template <typename> class d2>
void broadcast_rows(const dist_context ctx, const d1<r_t> row, const d2<r_t> mat, const bool discard = true) { MGGCN_DIST_LOOP(broadcast_rows(ctx[i], row[i], mat[i], discard)) }
template <typename r_t, template <typename> class dn_t>
void scale_mat(const dist_context ctx, const dn_t<r_t> mat, r_t s) { MGGCN_DIST_LOOP(scale_mat(ctx[i], mat[i], s)) }
#undef MGGCN_DIST_LOOP
