// dist_matrix.hpp -- the single-process, P-GPU types of the 1D row partition.
//
// Same names as the reference's src/dist_matrix.hpp: dist_context (:12-90),
// dist_row_csr_matrix (:170-260), dist_row_dn_matrix (:394-532), repl_dn_matrix (:534-639).
// One host thread drives P GPUs through per-GPU contexts; collectives go through
// libmggcn_comm.so (RCCL, include/mggcn_comm.h).  The column-partition classes
// (dist_csr_matrix, dist_dn_matrix) are dead code in the reference (only reachable from a
// commented-out branch of main.cpp) and are not rebuilt.
#pragma once

#include <cassert>
#include <numeric>
#include <ostream>
#include <vector>

#include "matrix.hpp"
#include "mggcn_comm.h"

class dist_context {
    std::vector<context> contexts;
    std::shared_ptr<mggcn_comm> comm_;

public:
    bool overlap = true;

    int bcast_stream_id() const { return overlap ? 1 : 0; }     // reference :20-22

    dist_context() = default;
    dist_context(std::size_t P, bool overlap = true) : overlap(overlap) {
        for (std::size_t i = 0; i < P; i++) contexts.emplace_back(i);
        comm_ = std::shared_ptr<mggcn_comm>(mggcn_comm_init_all((int)P, nullptr), &mggcn_comm_destroy);
    }

    auto size() const { return contexts.size(); }
    void sync() const { for (const auto &c : contexts) c.sync(); }
    const context &operator[](std::size_t i) const { return contexts[i]; }
    mggcn_comm *comm() const { return comm_.get(); }

    std::vector<mggcn_stream_t> streams(std::size_t stream_id) const {
        std::vector<mggcn_stream_t> s;
        for (const auto &c : contexts) s.push_back(c.stream(stream_id));
        return s;
    }

    void record(const std::string &name, std::size_t stream_id) const { for (const auto &c : contexts) c.record(name, stream_id); }
    void wait(const std::string &name, std::size_t stream_id) const { for (const auto &c : contexts) c.wait(name, stream_id); }
    void register_timer(const std::string &n, const std::string &b, const std::string &e) const {
        for (const auto &c : contexts) c.register_timer(n, b, e);
    }
    std::vector<float> measure(const std::string &name) const {
        std::vector<float> t;
        for (const auto &c : contexts) t.push_back(c.measure(name));
        return t;
    }
    // "<prefix><rank>_<name>:<ms>" (reference :86-89)
    void dump_timers(std::ostream &out, const std::string &prefix = "") const {
        for (std::size_t i = 0; i < size(); i++) contexts[i].dump_timers(out, prefix + std::to_string(i) + "_");
    }
};

// ---------------------------------------------------------------------------------------
// dist_row_csr_matrix: A cut into a P x P grid of CSR blocks, block (i,j) = rows of GPU i x
// rows-of-H of GPU j with block-local column indices (reference :215-259); in addition every
// row block keeps its "remote" part (all blocks but the diagonal one, merged, GLOBAL columns)
// for the all-gather schedule of ops.hpp.
// ---------------------------------------------------------------------------------------
template <typename x_t, typename v_t, typename r_t>
class dist_row_csr_matrix {
    using matrix_t = csr_matrix<x_t, v_t, r_t>;
    std::vector<std::vector<matrix_t>> As;
    std::vector<matrix_t> remote_;
    std::vector<v_t> p_;
    std::size_t M_ = 0;

    static std::vector<matrix_t> split_rows(const matrix_t &A, v_t rb, v_t re, const std::vector<v_t> &q) {
        const std::uint32_t nq = (std::uint32_t)q.size() - 1, rows = re - rb;
        std::vector<x_t> bip((std::size_t)nq * (rows + 1));
        mggcn_csr_block_split_count_host(A.indptr().data(), A.indices().data(), rb, re, q.data(), nq, bip.data());
        std::vector<std::vector<v_t>> idx(nq);
        std::vector<std::vector<r_t>> dat(nq);
        std::vector<v_t *> ip(nq);
        std::vector<r_t *> dp(nq);
        for (std::uint32_t j = 0; j < nq; j++) {
            const auto cnt = bip[(std::size_t)j * (rows + 1) + rows];
            idx[j].resize(cnt);
            dat[j].resize(cnt);
            ip[j] = idx[j].data();
            dp[j] = dat[j].data();
        }
        mggcn_csr_block_split_fill_host(A.indptr().data(), A.indices().data(), A.data().data(), rb, re, q.data(), nq,
                                        bip.data(), ip.data(), dp.data());
        std::vector<matrix_t> out;
        for (std::uint32_t j = 0; j < nq; j++) {
            std::vector<x_t> ptr(bip.begin() + (std::size_t)j * (rows + 1), bip.begin() + (std::size_t)(j + 1) * (rows + 1));
            out.emplace_back(std::move(ptr), std::move(idx[j]), std::move(dat[j]), q[j + 1] - q[j]);
        }
        return out;
    }

public:
    dist_row_csr_matrix() = default;

    dist_row_csr_matrix(const dist_context &, const matrix_t A, const std::vector<v_t> p, const std::vector<v_t> q)
        : p_(p), M_(A.m()) {
        assert(p == q);                                   // the only use in the reference (src/main.cpp:148-149)
        for (std::size_t i = 0; i + 1 < p.size(); i++) {
            As.push_back(split_rows(A, p[i], p[i + 1], q));
            // remote part: cut at [0, p_i, p_{i+1}, n], merge the two outer pieces with global columns
            const auto three = split_rows(A, p[i], p[i + 1], std::vector<v_t>{0, p[i], p[i + 1], (v_t)A.m()});
            const auto &L = three[0], &R = three[2];
            const v_t rows = p[i + 1] - p[i];
            std::vector<x_t> ptr(rows + 1, 0);
            std::vector<v_t> idx;
            std::vector<r_t> dat;
            idx.reserve(L.nnz() + R.nnz());
            dat.reserve(L.nnz() + R.nnz());
            for (v_t r = 0; r < rows; r++) {
                for (auto e = L.begin(r); e < L.end(r); e++) { idx.push_back(L.indices()[e]); dat.push_back(L.data()[e]); }
                for (auto e = R.begin(r); e < R.end(r); e++) { idx.push_back(R.indices()[e] + p[i + 1]); dat.push_back(R.data()[e]); }
                ptr[r + 1] = (x_t)idx.size();
            }
            remote_.emplace_back(std::move(ptr), std::move(idx), std::move(dat), (v_t)A.m());
        }
    }

    auto n() const { std::size_t N = 0; for (const auto &row : As) N += row[0].n(); return N; }
    auto m() const { return M_; }
    auto size() const { return As.size(); }
    auto operator[](std::pair<std::size_t, std::size_t> ij) const { return As[ij.first][ij.second]; }
    const matrix_t &remote(std::size_t i) const { return remote_[i]; }
    const std::vector<v_t> &bounds() const { return p_; }
};

// ---------------------------------------------------------------------------------------
// dist_row_dn_matrix: rows split evenly over the GPUs (reference :394-532)
// ---------------------------------------------------------------------------------------
template <typename r_t>
class dist_row_dn_matrix {
    using matrix_t = dn_matrix<r_t>;
    std::vector<matrix_t> As;

public:
    dist_row_dn_matrix() = default;

    dist_row_dn_matrix(const dist_context &ctx, std::size_t N, std::size_t M, std::vector<mggcn::device_ptr<r_t>> buffer = {}) {
        if (N % ctx.size() != 0) throw matrix_error("dist_row_dn_matrix: N % P != 0");     // assert at :428
        for (std::size_t i = 0; i < ctx.size(); i++) {
            ctx[i].set();
            As.emplace_back(N / ctx.size(), M, i < buffer.size() ? buffer[i] : nullptr);
        }
    }
    dist_row_dn_matrix(const dist_context &ctx, std::pair<std::size_t, std::size_t> shape) : dist_row_dn_matrix(ctx, shape.first, shape.second) {}

    // contiguous row slices of a host-readable matrix (reference :440-447)
    dist_row_dn_matrix(const dist_context &ctx, const matrix_t &A) : dist_row_dn_matrix(ctx, A.n(), A.m()) {
        const auto host = A.to_host();
        std::size_t off = 0;
        for (std::size_t i = 0; i < ctx.size(); i++) {
            ctx[i].set();
            mggcn::upload(As[i].buffer(), host.data() + off, As[i].size());
            off += As[i].size();
        }
    }

    auto n() const { std::size_t N = 0; for (const auto &A : As) N += A.n(); return N; }
    auto m() const { return As[0].m(); }
    auto shape() const { return std::make_pair(n(), m()); }
    auto size() const { return As.size(); }
    const matrix_t &operator[](std::size_t i) const { return As[i]; }

    // shard i to every GPU's bAs[j] (reference :458-467: group of P ncclBroadcast)
    void bcast(const dist_context &ctx, std::size_t i, const dist_row_dn_matrix &bAs, int stream_id = 1) const {
        std::vector<float *> recv;
        for (std::size_t j = 0; j < size(); j++) recv.push_back(bAs[j].buffer());
        const auto streams = ctx.streams(stream_id);
        mggcn_comm_broadcast_f32(ctx.comm(), As[i].buffer(), recv.data(), As[i].size(), (int)i, streams.data());
    }

    // all shards to every GPU's gathered[j] ([n x m]): the MI355X-first exchange
    void allgather(const dist_context &ctx, const std::vector<matrix_t> &gathered, int stream_id = 1) const {
        std::vector<const float *> send;
        std::vector<float *> recv;
        for (std::size_t j = 0; j < size(); j++) { send.push_back(As[j].buffer()); recv.push_back(gathered[j].buffer()); }
        const auto streams = ctx.streams(stream_id);
        mggcn_comm_allgather_f32(ctx.comm(), send.data(), recv.data(), As[0].size(), streams.data());
    }

    void zero(const dist_context &ctx) const { for (std::size_t i = 0; i < size(); i++) As[i].zero(ctx[i]); }

    void to_dn_matrix(const dist_context &ctx, std::vector<r_t> &host) const {
        ctx.sync();
        host.clear();
        for (std::size_t i = 0; i < size(); i++) {
            ctx[i].set();
            const auto h = As[i].to_host();
            host.insert(host.end(), h.begin(), h.end());
        }
    }
};

// ---------------------------------------------------------------------------------------
// repl_dn_matrix: a full copy on every GPU (reference :534-639)
// ---------------------------------------------------------------------------------------
template <typename r_t>
class repl_dn_matrix {
    using matrix_t = dn_matrix<r_t>;
    std::vector<matrix_t> As;

public:
    repl_dn_matrix() = default;
    repl_dn_matrix(const dist_context &ctx, std::size_t N, std::size_t M) {
        for (std::size_t i = 0; i < ctx.size(); i++) { ctx[i].set(); As.emplace_back(N, M); }
    }
    repl_dn_matrix(const dist_context &ctx, std::pair<std::size_t, std::size_t> s) : repl_dn_matrix(ctx, s.first, s.second) {}

    auto n() const { return As[0].n(); }
    auto m() const { return As[0].m(); }
    auto shape() const { return std::make_pair(n(), m()); }
    auto size() const { return As.size(); }
    const matrix_t &operator[](std::size_t i) const { return As[i]; }

    // in-place sum over the GPUs (reference :587-592)
    void allreduce(const dist_context &ctx) const {
        std::vector<float *> bufs;
        for (const auto &A : As) bufs.push_back(A.buffer());
        const auto streams = ctx.streams(0);
        mggcn_comm_allreduce_sum_f32(ctx.comm(), bufs.data(), As[0].size(), streams.data());
    }

    // the reference initialises GPU 0 and broadcasts (:601-609); the seed-99 host generator
    // gives every GPU the same bits without traffic
    void init(const dist_context &ctx, r_t gain = (r_t)std::sqrt(2 / (1 + 0.01 * 0.01))) {
        for (std::size_t i = 0; i < As.size(); i++) { ctx[i].set(); As[i].init(gain); }
    }
    void fill(const dist_context &ctx, r_t v) {
        for (std::size_t i = 0; i < As.size(); i++) { ctx[i].set(); As[i].fill(v); }
    }
    void zero(const dist_context &ctx) const { for (std::size_t i = 0; i < As.size(); i++) As[i].zero(ctx[i]); }
};
