// ops.hpp -- the reference's operator overload set (src/cuda_utils.hpp) over the C ABI.
//
// Same free-function names and argument order: matmul / get_matmul_buffer (SpMM :27-32,
// :94-102; GEMM :158-172; distributed SpMM :47-92; distributed row GEMMs :304-324), axpy,
// scale_mat, abssum, and the kernel wrappers (:470-748).  Each wrapper enqueues one C-ABI
// call on the context's compute stream.  Shape preconditions (assert in the reference)
// throw std::invalid_argument.
#pragma once

#include <array>
#include <memory>
#include <string>
#include <vector>

#include "dist_matrix.hpp"
#include "matrix.hpp"

inline void mggcn_require(bool ok, const char *what) {
    if (!ok) throw std::invalid_argument(what);
}

// ---- SpMM ------------------------------------------------------------------------------
template <typename x_t, typename v_t, typename r_t>
spmm_buffer get_matmul_buffer(const context ctx, const csr_matrix<x_t, v_t, r_t> A, const dn_matrix<r_t> B,
                              const dn_matrix<r_t> C, const r_t = 1, const r_t = 0) {
    mggcn_require(A.m() == B.n() && A.n() == C.n() && B.m() == C.m(), "get_matmul_buffer: shape mismatch");
    ctx.set();
    return A.plan(B.m());           // built once per (matrix, device, form) and shared, see csr_matrix::plan
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

// G_out = (alpha G . op(W)) .* leaky_relu'(Z): the next leaky_relu_backward folded into the GEMM epilogue
// (include/mggcn.h: mggcn_gemm_lrelu_bwd_f32; reference src/gcn.hpp:135-137 + :462-468)
template <typename r_t>
void matmul_lrelu_backward(const context ctx, const dn_matrix<r_t> A, const dn_matrix<r_t> B, const dn_matrix<r_t> Z,
                           const dn_matrix<r_t> C, const r_t alpha, const bool A_T = false, const bool B_T = false,
                           const r_t slope = 0.01f) {
    auto A_n = A.n(), A_m = A.m(), B_n = B.n(), B_m = B.m();
    if (A_T) std::swap(A_n, A_m);
    if (B_T) std::swap(B_n, B_m);
    mggcn_require(A_m == B_n && A_n == C.n() && B_m == C.m() && Z.shape() == C.shape(), "matmul_lrelu_backward: shape mismatch");
    ctx.set();
    const auto ws = mggcn_gemm_workspace_bytes(A_T, B_T, (uint32_t)A_n, (uint32_t)B_m, (uint32_t)A_m);
    mggcn_gemm_lrelu_bwd_f32(ctx.stream(0), A_T, B_T, (uint32_t)A_n, (uint32_t)B_m, (uint32_t)A_m, alpha, A.buffer(), A.m(),
                             B.buffer(), B.m(), Z.buffer(), Z.m(), slope, C.buffer(), C.m(), ctx.gemm_workspace(ws), ws);
}

// G_W = X^T G and G_b = 1^T G in one pass over G (include/mggcn.h: mggcn_gemm_tn_colsum_f32; reference src/gcn.hpp:125-134)
template <typename r_t>
void linear_backward_weights(const context ctx, const dn_matrix<r_t> X, const dn_matrix<r_t> G, const dn_matrix<r_t> G_W,
                             const dn_matrix<r_t> G_b) {
    mggcn_require(X.n() == G.n() && G_W.n() == X.m() && G_W.m() == G.m() && G_b.n() == 1 && G_b.m() == G.m(),
                  "linear_backward_weights: shape mismatch");
    ctx.set();
    const auto ws = mggcn_gemm_tn_colsum_workspace_bytes((uint32_t)X.m(), (uint32_t)G.m(), (uint32_t)X.n());
    mggcn_gemm_tn_colsum_f32(ctx.stream(0), (uint32_t)X.m(), (uint32_t)G.m(), (uint32_t)X.n(), (r_t)1, X.buffer(), X.m(), G.buffer(),
                             G.m(), G_W.buffer(), G_W.m(), G_b.buffer(), ctx.gemm_workspace(ws), ws);
}

// every (param, grad, m, v) quadruple of a model on one GPU as a device table: ONE Adam launch per epoch
// (include/mggcn.h: mggcn_adam_multi_f32; reference src/gcn.hpp:146-172 runs 7 launches per layer)
template <typename r_t>
class adam_table {
    mggcn::device_ptr<mggcn_adam_tensor> dev_;
    std::vector<dn_matrix<r_t>> keep_;          // the table holds raw pointers
    uint32_t n_ = 0, blocks_ = 0;

public:
    adam_table() = default;
    // entries: {param, grad, m, v} + weight decay of each
    adam_table(const context &ctx, const std::vector<std::array<dn_matrix<r_t>, 4>> &tensors, const std::vector<r_t> &wd) {
        std::vector<mggcn_adam_tensor> tab;
        for (std::size_t k = 0; k < tensors.size(); k++) {
            const auto &t = tensors[k];
            tab.push_back({t[0].buffer(), t[1].buffer(), t[2].buffer(), t[3].buffer(), (uint64_t)t[0].size(), wd[k], blocks_});
            blocks_ += mggcn_adam_multi_blocks(t[0].size());
            for (const auto &x : t) keep_.push_back(x);
        }
        n_ = (uint32_t)tab.size();
        ctx.set();
        dev_ = mggcn::device_malloc<mggcn_adam_tensor>(std::max<std::size_t>(tab.size(), 1));
        if (!tab.empty()) mggcn::upload(dev_.get(), tab.data(), tab.size());
    }
    explicit operator bool() const { return n_ != 0; }
    void step(const context &ctx, r_t lr, r_t b1, r_t b2, r_t c1, r_t c2, r_t eps) const {
        ctx.set();
        mggcn_adam_multi_f32(ctx.stream(0), dev_.get(), n_, blocks_, lr, b1, b2, c1, c2, eps);
    }
};

// ---- distributed SpMM: C_j = beta C_j + alpha sum_i A[j,i] B_i ------------------------------
// Three schedules of the same sum (MGGCN_DIST_MODE / dist_gcn's `mode`):
//   allgather  K all-gathers of one piece of every shard each, queued back to back on the comm stream;
//              diagonal block first (no dependency), then the K pieces of the merged remote block as
//              they land -- the SpMM over piece c runs while piece c+1 is on the wire         (default)
//   halo       every GPU packs the rows its peers' blocks reference, ONE variable-size exchange, the
//              remote block renumbered to the receive layout                     
//   rounds     the reference's P broadcast rounds, double-buffered            (src/cuda_utils.hpp:57-92)
enum class dist_mode { allgather, rounds, halo };

inline dist_mode dist_mode_from_string(const std::string &s) {
    if (s == "rounds") return dist_mode::rounds;
    if (s == "halo") return dist_mode::halo;
    if (s.empty() || s == "allgather") return dist_mode::allgather;
    throw std::invalid_argument("unknown distributed schedule '" + s + "' (allgather | halo | rounds)");
}

struct dist_spmm_buffers {
    std::vector<std::vector<spmm_buffer>> block;   // [j][i]  rounds: every block; otherwise only the diagonal [j][j]
    std::vector<std::vector<spmm_buffer>> piece;   // [j][c]  all-gather schedule: pieces of the merged remote block
    std::vector<spmm_buffer> halo;                 // [j]     halo schedule: remote block in receive layout
};

template <typename x_t, typename v_t, typename r_t>
dist_spmm_buffers get_matmul_buffer(const dist_context ctx, const dist_row_csr_matrix<x_t, v_t, r_t> A,
                                    const dist_row_dn_matrix<r_t> B, const dist_row_dn_matrix<r_t> C, dist_mode mode) {
    dist_spmm_buffers out;
    const auto P = ctx.size();
    ctx.drain();        // plans and device copies of the blocks are created by the calling thread: no command of an
                        // enqueue thread may be using the same matrices meanwhile (first epoch only)
    out.block.resize(P);
    out.piece.resize(P);
    // these SpMMs run while the exchange's kernels (RCCL channels, the peer-copy transport's sums) share the device: their launch
    // rounds leave (at least) 12 CUs' worth of wave slots free (include/mggcn.h: mggcn_spmm_plan_reserved_cus)
    struct reserve_scope {
        bool on;
        explicit reserve_scope(bool on) : on(on) { if (on) mggcn_spmm_plan_reserved_cus(12); }
        ~reserve_scope() { if (on) mggcn_spmm_plan_reserved_cus(0); }
    } reserve(P > 1 && ctx.overlap);
    {   // every rank's plans of this width, built side by side (csr_matrix::prebuild_plans), then picked up below
        std::vector<typename csr_matrix<x_t, v_t, r_t>::plan_want> wants;
        for (std::size_t j = 0; j < P; j++) {
            const int dev = (int)ctx[j].device();
            for (std::size_t i = 0; i < P; i++)
                if (mode == dist_mode::rounds || i == j) wants.push_back({A[{j, i}], B.m(), dev});
            if (mode == dist_mode::allgather && P > 1)
                for (std::size_t c = 0; c < A.chunks(); c++) wants.push_back({A.remote_chunk(j, c), B.m(), dev});
            if (mode == dist_mode::halo && P > 1) wants.push_back({A.halo_remote(j), B.m(), dev});
        }
        csr_matrix<x_t, v_t, r_t>::prebuild_plans(wants);
    }
    for (std::size_t j = 0; j < P; j++) {
        ctx[j].set();
        out.block[j].resize(P);
        for (std::size_t i = 0; i < P; i++)
            if (mode == dist_mode::rounds || i == j) out.block[j][i] = A[{j, i}].plan(B.m());
        if (mode == dist_mode::allgather && P > 1)
            for (std::size_t c = 0; c < A.chunks(); c++) out.piece[j].push_back(A.remote_chunk(j, c).plan(B.m()));
        if (mode == dist_mode::halo) out.halo.push_back(P > 1 ? A.halo_remote(j).plan(B.m()) : spmm_buffer());
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
    const int cs = ctx.bcast_stream_id();
    ctx.record(name + "0_matmul-spmm", 0);
    ctx.wait(name + "0_matmul-spmm", cs);
    for (std::size_t i = 0; i < P; i++) {
        if (i > 1) ctx.wait(name + std::to_string(i - 1) + "_matmul-spmm", cs);   // double-buffer hazard (:66-67)
        ctx.record(name + std::to_string(i) + "_matmul-bcast-start", cs);
        B.bcast(ctx, i, B_bcast[i % 2], cs);
        ctx.record(name + std::to_string(i) + "_matmul-bcast-finish", cs);
        ctx.wait(name + std::to_string(i) + "_matmul-bcast-finish", 0);
        for (std::size_t j = 0; j < P; j++)
            ctx.on(j, [c = ctx[j], a = A[{j, i}], b = B_bcast[i % 2][j], cc = C[j], pl = ext.block[j][i], alpha,
                       bt = i == 0 ? beta : (r_t)1, fl = i + 1 == P ? last_flags : 0u] { matmul(c, a, b, cc, pl, alpha, bt, fl); });
        if (i + 1 == P) ctx.release_sends(0);             // B may be overwritten once every GPU has read its broadcasts
        ctx.record(name + std::to_string(i + 1) + "_matmul-spmm", 0);
    }
    ctx.register_timer(name + "matmul-spmm", name + "0_matmul-spmm", name + std::to_string(P) + "_matmul-spmm");
}

// MI355X-first schedule.  gathered[j]: resident [n x d] receive buffer of GPU j; piece c occupies rows
// P*cb[c] .. P*cb[c+1] of it (rank-major inside the piece), which is the column layout of A.remote_chunk(j, c).
template <typename x_t, typename v_t, typename r_t>
void matmul_allgather(dist_context ctx, dist_row_csr_matrix<x_t, v_t, r_t> A, dist_row_dn_matrix<r_t> B,
                      dist_row_dn_matrix<r_t> C, const dist_spmm_buffers &ext, const std::vector<dn_matrix<r_t>> &gathered,
                      const r_t alpha, const r_t beta, const std::string name = "", const uint32_t last_flags = 0) {
    const auto P = ctx.size();
    const int cs = ctx.bcast_stream_id();
    const auto &cb = A.chunk_bounds();
    const std::size_t K = A.chunks(), d = B.m();
    ctx.record(name + "0_matmul-spmm", 0);
    ctx.wait(name + "0_matmul-spmm", cs);                 // the comm stream sees the producer of B (and the
                                                          // previous call's readers of `gathered`)
    ctx.record(name + "0_matmul-bcast-start", cs);
    for (std::size_t c = 0; c < K; c++) {                 // all K pieces are queued at once and land in order
        // (one rank: nothing is remote and nobody reads the gathered copy -- the "all-gather" would be a copy kernel that shares
        // the device with the local SpMM for nothing)
        if (P > 1 || ctx.self_gather) B.allgather(ctx, gathered, cb[c], cb[c + 1], cs);
        ctx.record(name + std::to_string(c) + "_matmul-bcast-finish", cs);
    }
    for (std::size_t j = 0; j < P; j++)                   // local block: no dependency on the exchange
        ctx.on(j, [c = ctx[j], a = A[{j, j}], b = B[j], cc = C[j], pl = ext.block[j][j], alpha, beta,
                   fl = P == 1 ? last_flags : 0u] { matmul(c, a, b, cc, pl, alpha, beta, fl); });
    for (std::size_t c = 0; c < K; c++) {
        ctx.wait(name + std::to_string(c) + "_matmul-bcast-finish", 0);
        if (P == 1) continue;
        const std::size_t len = cb[c + 1] - cb[c];
        for (std::size_t j = 0; j < P; j++) {
            const dn_matrix<r_t> piece(P * len, d, mggcn::device_view(gathered[j].shared_buffer(), P * cb[c] * d));
            ctx.on(j, [cx = ctx[j], a = A.remote_chunk(j, c), piece, cc = C[j], pl = ext.piece[j][c], alpha,
                       fl = c + 1 == K ? last_flags : 0u] { matmul(cx, a, piece, cc, pl, alpha, (r_t)1, fl); });
        }
    }
    ctx.release_sends(0);                                 // a GPU's shard may be overwritten once its peers have pulled it
    ctx.record(name + "1_matmul-spmm", 0);
    ctx.register_timer(name + "matmul-spmm", name + "0_matmul-spmm", name + "1_matmul-spmm");
}

// Halo schedule: what each GPU sends is fixed by the partition, so the index lists live with the caller
// (dist_halo_plan, built once per matrix); recv[j] holds [need(j,0) | need(j,1) | ...] rows of width d.
template <typename r_t>
struct dist_halo_plan {
    std::vector<mggcn::device_ptr<std::uint32_t>> send_idx;   // [j]: rows of shard j to pack, destination order
    std::vector<std::size_t> send_rows, recv_rows;            // [j]
    std::vector<std::size_t> rows;                            // [j*P + k]: rows GPU j sends to GPU k
    std::vector<mggcn::device_ptr<r_t>> send_buf;             // [j], grown on demand
    std::size_t send_width = 0;

    dist_halo_plan() = default;
    template <typename x_t, typename v_t>
    dist_halo_plan(const dist_context &ctx, const dist_row_csr_matrix<x_t, v_t, r_t> &A) {
        const auto P = ctx.size();
        rows.assign(P * P, 0);
        send_rows.assign(P, 0);
        recv_rows.assign(P, 0);
        for (std::size_t j = 0; j < P; j++) {
            std::vector<std::uint32_t> idx;
            for (std::size_t k = 0; k < P; k++) {
                const auto &nd = A.halo_need(k, j);           // rows of shard j that GPU k needs
                rows[j * P + k] = nd.size();
                idx.insert(idx.end(), nd.begin(), nd.end());
                recv_rows[k] += nd.size();
            }
            send_rows[j] = idx.size();
            ctx[j].set();
            send_idx.push_back(mggcn::device_malloc<std::uint32_t>(std::max<std::size_t>(idx.size(), 1)));
            if (!idx.empty()) mggcn::upload(send_idx.back().get(), idx.data(), idx.size());
        }
        send_buf.resize(P);
    }
    void reserve(const dist_context &ctx, std::size_t d) {
        if (d <= send_width) return;
        ctx.sync();
        for (std::size_t j = 0; j < ctx.size(); j++) {
            ctx[j].set();
            send_buf[j] = mggcn::device_malloc<r_t>(std::max<std::size_t>(send_rows[j], 1) * d);
        }
        send_width = d;
    }
};

template <typename x_t, typename v_t, typename r_t>
void matmul_halo(dist_context ctx, dist_row_csr_matrix<x_t, v_t, r_t> A, dist_row_dn_matrix<r_t> B, dist_row_dn_matrix<r_t> C,
                 const dist_spmm_buffers &ext, dist_halo_plan<r_t> &halo, const std::vector<dn_matrix<r_t>> &recv,
                 const r_t alpha, const r_t beta, const std::string name = "", const uint32_t last_flags = 0) {
    const auto P = ctx.size();
    const int cs = ctx.bcast_stream_id();
    const std::size_t d = B.m();
    halo.reserve(ctx, d);
    ctx.record(name + "0_matmul-spmm", 0);
    for (std::size_t j = 0; j < P; j++)                   // pack on the compute stream
        if (halo.send_rows[j])
            ctx.on(j, [c = ctx[j], b = B[j], idx = halo.send_idx[j], rows = halo.send_rows[j], d, out = halo.send_buf[j]] {
                c.set();
                mggcn_gather_rows_f32(c.stream(0), b.buffer(), d, idx.get(), rows, (uint32_t)d, out.get(), d);
            });
    ctx.record(name + "0_matmul-halo-packed", 0);
    ctx.wait(name + "0_matmul-halo-packed", cs);
    ctx.record(name + "0_matmul-bcast-start", cs);
    {
        struct exchange_args { std::vector<const float *> send; std::vector<float *> rcv; std::vector<std::size_t> counts;
                               std::vector<mggcn::device_ptr<r_t>> keep; };
        auto x = std::make_shared<exchange_args>();
        x->counts.resize(P * P);
        for (std::size_t j = 0; j < P; j++) { x->send.push_back(halo.send_buf[j].get()); x->rcv.push_back(recv[j].buffer()); x->keep.push_back(halo.send_buf[j]); }
        for (std::size_t q = 0; q < P * P; q++) x->counts[q] = halo.rows[q] * d;
        if (!ctx.threaded()) {
            const auto streams = ctx.streams(cs);
            mggcn_comm_alltoallv_f32(ctx.comm(), x->send.data(), x->rcv.data(), x->counts.data(), streams.data());
        } else {
            for (std::size_t j = 0; j < P; j++)
                ctx.on(j, [cm = ctx.comm(), j, x, st = ctx[j].stream(cs)] {
                    mggcn_comm_alltoallv_rank_f32(cm, (int)j, x->send.data(), x->rcv.data(), x->counts.data(), st);
                });
        }
    }
    ctx.record(name + "0_matmul-bcast-finish", cs);
    for (std::size_t j = 0; j < P; j++)
        ctx.on(j, [c = ctx[j], a = A[{j, j}], b = B[j], cc = C[j], pl = ext.block[j][j], alpha, beta,
                   fl = P == 1 ? last_flags : 0u] { matmul(c, a, b, cc, pl, alpha, beta, fl); });
    ctx.wait(name + "0_matmul-bcast-finish", 0);
    if (P > 1)
        for (std::size_t j = 0; j < P; j++) {
            const dn_matrix<r_t> r(std::max<std::size_t>(halo.recv_rows[j], 1), d, recv[j].shared_buffer());
            ctx.on(j, [c = ctx[j], a = A.halo_remote(j), r, cc = C[j], pl = ext.halo[j], alpha, last_flags] {
                matmul(c, a, r, cc, pl, alpha, (r_t)1, last_flags);
            });
        }
    ctx.release_sends(0);                                 // the pack buffers are rewritten by the next call
    ctx.record(name + "1_matmul-spmm", 0);
    ctx.register_timer(name + "matmul-spmm", name + "0_matmul-spmm", name + "1_matmul-spmm");
}

// ---- distributed row GEMMs (reference src/cuda_utils.hpp:304-324) ---------------------------
template <typename r_t>
void matmul(const dist_context ctx, const dist_row_dn_matrix<r_t> A, const dist_row_dn_matrix<r_t> B,
            const repl_dn_matrix<r_t> C, const r_t alpha, const r_t beta) {      // C = sum_i A_i^T B_i
    for (std::size_t i = 0; i < ctx.size(); i++)
        ctx.on(i, [c = ctx[i], a = A[i], b = B[i], cc = C[i], alpha, beta] { matmul(c, a, b, cc, alpha, beta, true); });
    C.allreduce(ctx);
}

template <typename r_t>
void matmul(const dist_context ctx, const dist_row_dn_matrix<r_t> A, const repl_dn_matrix<r_t> B,
            const dist_row_dn_matrix<r_t> C, const r_t alpha, const r_t beta, const bool B_T = false) {
    for (std::size_t i = 0; i < ctx.size(); i++)
        ctx.on(i, [c = ctx[i], a = A[i], b = B[i], cc = C[i], alpha, beta, B_T] { matmul(c, a, b, cc, alpha, beta, false, B_T); });
}

template <typename r_t>
void linear_forward(const dist_context ctx, const dist_row_dn_matrix<r_t> X, const repl_dn_matrix<r_t> W,
                    const repl_dn_matrix<r_t> b, const dist_row_dn_matrix<r_t> XW) {
    for (std::size_t i = 0; i < ctx.size(); i++)
        ctx.on(i, [c = ctx[i], x = X[i], w = W[i], bb = b[i], xw = XW[i]] { linear_forward(c, x, w, bb, xw); });
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
// out of place: logits H -> gradient G (the loss layer's copy = true without the copy)
template <typename r_t, typename x_t>
void softmax_xent_fused(const context ctx, const dn_matrix<r_t> H, const dn_matrix<r_t> G, const dn_matrix<x_t> Y, r_t grad_scale,
                        r_t *sums_device) {
    mggcn_require(H.n() == Y.n() && Y.m() == 1, "softmax_xent_fused: labels must be n x 1");
    mggcn_require(G.n() == H.n() && G.m() == H.m(), "softmax_xent_fused: gradient matrix must have the logits' shape");
    ctx.set();
    mggcn_softmax_xent_fused_from_f32(ctx.stream(0), H.buffer(), G.buffer(), Y.buffer(), H.n(), H.m(), grad_scale, sums_device);
}

// dist_context forms: per-GPU loops, as in the reference's "template<dn_t>" overloads
template <typename r_t, template <typename> class dn_t>
void leaky_relu_forward(const dist_context ctx, const dn_t<r_t> in, const dn_t<r_t> out, r_t a = 0.01) {
    for (std::size_t i = 0; i < ctx.size(); i++) ctx.on(i, [c = ctx[i], x = in[i], y = out[i], a] { leaky_relu_forward(c, x, y, a); });
}
template <typename r_t, template <typename> class dn_t>
void leaky_relu_backward(const dist_context ctx, const dn_t<r_t> in, const dn_t<r_t> G_in, const dn_t<r_t> G_out, r_t a = 0.01) {
    for (std::size_t i = 0; i < ctx.size(); i++)
        ctx.on(i, [c = ctx[i], x = in[i], g = G_in[i], y = G_out[i], a] { leaky_relu_backward(c, x, g, y, a); });
}
template <typename r_t, template <typename> class d1, template <typename> class d2>
void broadcast_rows(const dist_context ctx, const d1<r_t> row, const d2<r_t> mat, const bool discard = true) {
    for (std::size_t i = 0; i < ctx.size(); i++) ctx.on(i, [c = ctx[i], r = row[i], m = mat[i], discard] { broadcast_rows(c, r, m, discard); });
}
template <typename r_t, template <typename> class dn_t>
void scale_mat(const dist_context ctx, const dn_t<r_t> mat, r_t s) {
    for (std::size_t i = 0; i < ctx.size(); i++) ctx.on(i, [c = ctx[i], m = mat[i], s] { scale_mat(c, m, s); });
}
