// dist_matrix.hpp -- the single-process, P-GPU types of the 1D row partition.
//
// Same names as the reference's src/dist_matrix.hpp: dist_context (:12-90),
// dist_row_csr_matrix (:170-260), dist_row_dn_matrix (:394-532), repl_dn_matrix (:534-639).
// One process drives P GPUs through per-GPU contexts -- by default with one enqueue thread per GPU
// (enqueue.hpp; the reference's loops `for j: ctx[j].set(); launch` push GPU j's body to thread j);
// collectives go through libmggcn_comm.so (include/mggcn_comm.h: RCCL, or event-ordered peer copies).
// The column-partition classes (dist_csr_matrix, dist_dn_matrix) are dead code in the reference (only
// reachable from a commented-out branch of main.cpp) and are not rebuilt.
#pragma once

#include <algorithm>
#include <cassert>
#include <chrono>
#include <cstdlib>
#include <numeric>
#include <ostream>
#include <mutex>
#include <thread>
#include <vector>

#include "enqueue.hpp"
#include "matrix.hpp"
#include "mggcn_comm.h"

class dist_context {
    std::vector<context> contexts;
    std::shared_ptr<mggcn_comm> comm_;
    // one enqueue thread per GPU (enqueue.hpp); null: the calling thread issues everything itself, GPU after GPU,
    // like the reference (src/cuda_utils.hpp:57-92).  Shared by the copies of a context (contexts are passed by value).
    std::shared_ptr<mggcn::enqueue_pool> pool_;
    // host seconds sync() has spent waiting for the DEVICES (after every command had been issued): an epoch's wall time
    // minus this is what the host needed to issue it -- the figure that says whether the process is host-bound
    std::shared_ptr<double> device_wait_s_ = std::make_shared<double>(0.0);

public:
    bool overlap = true;
    // MGGCN_DIST_SELF_GATHER=1 (tests of the transport): the all-gather schedule exchanges with ONE rank too (RCCL's
    // single-rank all-gather); by default one rank exchanges nothing
    bool self_gather = false;

    int bcast_stream_id() const { return overlap ? 1 : 0; }     // reference :20-22

    dist_context() = default;
    // Rank i drives GPU i (reference :26-31).  MGGCN_OVERSUBSCRIBE=1 wraps the ranks over the GPUs that
    // are visible (rank i -> GPU i mod count): the whole P-rank schedule on fewer GPUs, with the
    // communication library's peer-copy transport -- a rehearsal / debugging aid, not a fast path.
    // MGGCN_ENQUEUE_THREADS=0|1 (default: 1 from two ranks on): one enqueue thread per GPU.
    dist_context(std::size_t P, bool overlap = true) : overlap(overlap) {
        const int count = mggcn_device_count();
        const char *os = std::getenv("MGGCN_OVERSUBSCRIBE");
        const bool wrap = os && std::atoi(os) != 0;
        if ((int)P > count && !wrap) throw matrix_error("dist_context: fewer GPUs visible than ranks");
        std::vector<int> devices;
        for (std::size_t i = 0; i < P; i++) devices.push_back((int)(i % (std::size_t)std::max(count, 1)));
        for (std::size_t i = 0; i < P; i++) contexts.emplace_back((std::size_t)devices[i]);
        comm_ = std::shared_ptr<mggcn_comm>(mggcn_comm_init_all((int)P, devices.data()), &mggcn_comm_destroy);
        // the host layer releases its send buffers itself, on the compute stream at the end of an SpMM (ops.hpp), and
        // its remote blocks never read a rank's own piece of the gathered matrix
        mggcn_comm_set_exchange_flags(comm_.get(), MGGCN_COMM_DEFER_RELEASE | MGGCN_COMM_SKIP_SELF);
        if (const char *sg = std::getenv("MGGCN_DIST_SELF_GATHER")) self_gather = std::atoi(sg) != 0;
        const char *et = std::getenv("MGGCN_ENQUEUE_THREADS");
        if (et ? std::atoi(et) != 0 : P > 1)
            pool_ = std::make_shared<mggcn::enqueue_pool>(P, [devices](std::size_t j) { mggcn_set_device(devices[j]); });
    }

    auto size() const { return contexts.size(); }
    const context &operator[](std::size_t i) const { return contexts[i]; }
    mggcn_comm *comm() const { return comm_.get(); }
    std::string transport() const { return mggcn_comm_transport(comm_.get()); }

    // ---- per-GPU command queues ---------------------------------------------------------------
    bool threaded() const { return (bool)pool_; }
    // GPU j's next command: runs f() on rank j's enqueue thread, after everything pushed for j before (or right here
    // when the context has no threads).  f captures BY VALUE and never holds a dist_context (enqueue.hpp).
    template <typename F>
    void on(std::size_t j, F &&f) const {
        if (pool_) pool_->push(j, std::forward<F>(f));
        else f();
    }
    // every command pushed so far has been issued; rethrows what a command threw
    void drain() const { if (pool_) pool_->drain(); }
    void sync() const {
        drain();
        const auto t0 = std::chrono::steady_clock::now();
        for (const auto &c : contexts) c.sync();
        *device_wait_s_ += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    double device_wait_seconds() const { return *device_wait_s_; }

    std::vector<mggcn_stream_t> streams(std::size_t stream_id) const {
        std::vector<mggcn_stream_t> s;
        for (const auto &c : contexts) s.push_back(c.stream(stream_id));
        return s;
    }

    void record(const std::string &name, std::size_t stream_id) const {
        for (std::size_t j = 0; j < contexts.size(); j++) on(j, [c = contexts[j], name, stream_id] { c.record(name, stream_id); });
    }
    void wait(const std::string &name, std::size_t stream_id) const {
        for (std::size_t j = 0; j < contexts.size(); j++) on(j, [c = contexts[j], name, stream_id] { c.wait(name, stream_id); });
    }
    void register_timer(const std::string &n, const std::string &b, const std::string &e) const {
        for (std::size_t j = 0; j < contexts.size(); j++) on(j, [c = contexts[j], n, b, e] { c.register_timer(n, b, e); });
    }
    // stream `stream_id` of every GPU waits until its peers have read what the GPU sent in the exchanges so far
    // (include/mggcn_comm.h: MGGCN_COMM_DEFER_RELEASE; nothing to do on the rccl transport)
    void release_sends(std::size_t stream_id) const {
        if (!pool_) { const auto s = streams(stream_id); mggcn_comm_release(comm_.get(), s.data()); return; }
        for (std::size_t j = 0; j < contexts.size(); j++)
            on(j, [cm = comm_, j, st = contexts[j].stream(stream_id)] { mggcn_comm_release_rank(cm.get(), (int)j, st); });
    }
    std::vector<float> measure(const std::string &name) const {
        drain();
        std::vector<float> t;
        for (const auto &c : contexts) t.push_back(c.measure(name));
        return t;
    }
    // "<prefix><rank>_<name>:<ms>" (reference :86-89)
    void dump_timers(std::ostream &out, const std::string &prefix = "") const {
        drain();
        for (std::size_t i = 0; i < size(); i++) contexts[i].dump_timers(out, prefix + std::to_string(i) + "_");
    }
};

// pieces the exchange of one SpMM is cut into (all-gather schedule): with K pieces the SpMM over
// piece c runs while piece c+1 is on the wire.  2 at P = 2, 4 above; MGGCN_DIST_CHUNKS overrides.
inline std::size_t mggcn_default_chunks(std::size_t P) {
    if (const char *s = std::getenv("MGGCN_DIST_CHUNKS")) return std::max<std::size_t>(1, std::strtoull(s, nullptr, 10));
    return P <= 1 ? 1 : (P == 2 ? 2 : 4);
}

// ---------------------------------------------------------------------------------------
// dist_row_csr_matrix: A cut into a P x P grid of CSR blocks, block (i,j) = rows of GPU i x
// rows-of-H of GPU j with block-local column indices (reference :215-259).  For the MI355X
// schedules of ops.hpp every row block additionally keeps
//   * its "remote" part (all blocks but the diagonal one, merged) cut into K pieces by the
//     piece of the source shard a column lives in, columns renumbered to the layout one
//     all-gather of that piece produces                               (all-gather schedule)
//   * the halo form: need(i,s) = distinct rows of shard s that row block i references, and the
//     remote part renumbered to the receive layout [need(i,0) | need(i,1) | ...]   (halo schedule)
// ---------------------------------------------------------------------------------------
template <typename x_t, typename v_t, typename r_t>
class dist_row_csr_matrix {
    using matrix_t = csr_matrix<x_t, v_t, r_t>;
    // Shared by the copies of one matrix (the classes are passed by value like the reference's): the block split is
    // built by the constructor, the two MI355X-first forms on first use -- a model runs ONE schedule, and at the
    // Reddit shape the halo lists alone (sort + unique + a binary search per non-zero) cost more host time than the
    // rest of the partition.  Every builder runs one host thread per row block (the reference's constructor is one
    // serial loop, src/dist_matrix.hpp:215-259; at P = 8 that was 20 s of single-thread work before the first epoch).
    struct state {
        std::vector<std::vector<matrix_t>> As;                 // [i][s]
        std::vector<std::vector<matrix_t>> chunks;             // [i][c]      all-gather schedule, built on first use
        std::vector<matrix_t> halo_remote;                     // [i]         halo schedule, built on first use
        std::vector<std::vector<std::vector<v_t>>> need;       // [i][s]
        std::vector<v_t> chunk_bounds;                         // K + 1 row offsets inside a shard
        std::once_flag chunks_once, halo_once;
    };
    std::shared_ptr<state> st_ = std::make_shared<state>();
    std::vector<v_t> p_;
    std::size_t M_ = 0;

    template <typename F>
    static void for_each_block(std::size_t P, F &&fn) {
        if (P <= 1) { for (std::size_t i = 0; i < P; i++) fn(i); return; }
        std::vector<std::thread> th;
        for (std::size_t i = 0; i < P; i++) th.emplace_back(fn, i);
        for (auto &t : th) t.join();
    }

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

    // The off-diagonal blocks of row block i merged into ONE CSR (row order and, inside a row, source-rank
    // order then original order kept), entry (block s, local column l) kept when keep(l) and given column col(s, l).
    template <typename K, typename F>
    static matrix_t merge_remote(const std::vector<matrix_t> &blocks, std::size_t i, v_t n_cols, K &&keep, F &&col) {
        const v_t rows = blocks[0].n();
        std::vector<x_t> ptr(rows + 1, 0);
        std::size_t total = 0;                                         // exact count first: no reallocation, no masked copies
        for (std::size_t s = 0; s < blocks.size(); s++) {
            if (s == i) continue;
            const auto &b = blocks[s];
            for (v_t r = 0; r < rows; r++)
                for (auto e = b.begin(r); e < b.end(r); e++)
                    if (keep(b.indices()[e])) { ptr[r + 1]++; total++; }
        }
        for (v_t r = 0; r < rows; r++) ptr[r + 1] += ptr[r];
        std::vector<v_t> idx(total);
        std::vector<r_t> dat(total);
        std::vector<x_t> cur(ptr.begin(), ptr.end() - 1);
        for (std::size_t s = 0; s < blocks.size(); s++) {            // source-rank order inside a row: s ascending
            if (s == i) continue;
            const auto &b = blocks[s];
            for (v_t r = 0; r < rows; r++)
                for (auto e = b.begin(r); e < b.end(r); e++) {
                    const v_t l = b.indices()[e];
                    if (!keep(l)) continue;
                    const x_t at = cur[r]++;
                    idx[at] = col(s, l);
                    dat[at] = b.data()[e];
                }
        }
        return matrix_t(std::move(ptr), std::move(idx), std::move(dat), n_cols);
    }

    // (a) all-gather schedule: piece c gathers rows cb[c]..cb[c+1] of EVERY shard in rank order, so
    //     entry (s, l) of piece c = piece_of[l] lands at column s * len_c + (l - cb[c])
    void build_chunks() const {
        auto &S = *st_;
        const std::size_t P = S.As.size(), K = S.chunk_bounds.size() - 1;
        S.chunks.assign(P, {});
        for_each_block(P, [&](std::size_t i) {
            const auto &blocks = S.As[i];
            for (std::size_t c = 0; c < K; c++) {
                const v_t lo = S.chunk_bounds[c], hi = S.chunk_bounds[c + 1], len = hi - lo;
                S.chunks[i].push_back(merge_remote(blocks, i, (v_t)(P * len), [&](v_t l) { return l >= lo && l < hi; },
                                                   [&](std::size_t s, v_t l) { return (v_t)(s * len + (l - lo)); }));
            }
        });
    }

    // (b) halo schedule: distinct referenced rows per source shard, receive layout in source order
    void build_halo() const {
        auto &S = *st_;
        const std::size_t P = S.As.size();
        S.need.assign(P, std::vector<std::vector<v_t>>(P));
        S.halo_remote.assign(P, matrix_t());
        for_each_block(P, [&](std::size_t i) {
            const auto &blocks = S.As[i];
            const v_t shard = blocks[0].n();
            std::vector<v_t> off(P + 1, 0);
            // position of every referenced local row in its source's need list: one flag pass, no sort, no search
            std::vector<std::vector<v_t>> pos(P);
            for (std::size_t s = 0; s < P; s++) {
                if (s != i) {
                    const auto width = blocks[s].m();
                    std::vector<unsigned char> used(width, 0);
                    for (const auto l : blocks[s].indices()) used[l] = 1;
                    pos[s].assign(width, 0);
                    auto &nd = S.need[i][s];
                    for (v_t l = 0; l < width; l++)
                        if (used[l]) { pos[s][l] = (v_t)nd.size(); nd.push_back(l); }
                }
                off[s + 1] = off[s] + (v_t)S.need[i][s].size();
            }
            (void)shard;
            S.halo_remote[i] = merge_remote(blocks, i, std::max<v_t>(off[P], 1), [](v_t) { return true; },
                                            [&](std::size_t s, v_t l) { return (v_t)(off[s] + pos[s][l]); });
        });
    }
    void need_chunks() const { std::call_once(st_->chunks_once, [this] { build_chunks(); }); }
    void need_halo() const { std::call_once(st_->halo_once, [this] { build_halo(); }); }

public:
    dist_row_csr_matrix() = default;

    dist_row_csr_matrix(const dist_context &, const matrix_t A, const std::vector<v_t> p, const std::vector<v_t> q,
                        std::size_t chunks = 0)
        : p_(p), M_(A.m()) {
        assert(p == q);                                   // the only use in the reference (src/main.cpp:148-149)
        const std::size_t P = p.size() - 1;
        const v_t rows = p[1] - p[0];                     // equal shards (N % P == 0, reference :428)
        std::size_t K = chunks ? chunks : mggcn_default_chunks(P);
        K = std::max<std::size_t>(1, std::min<std::size_t>(K, std::max<v_t>(rows, 1)));
        for (std::size_t c = 0; c <= K; c++) st_->chunk_bounds.push_back((v_t)(c * (std::size_t)rows / K));
        st_->As.resize(P);
        for_each_block(P, [&](std::size_t i) { st_->As[i] = split_rows(A, p[i], p[i + 1], q); });
    }

    auto n() const { std::size_t N = 0; for (const auto &row : st_->As) N += row[0].n(); return N; }
    auto m() const { return M_; }
    auto size() const { return st_->As.size(); }
    auto operator[](std::pair<std::size_t, std::size_t> ij) const { return st_->As[ij.first][ij.second]; }
    const std::vector<v_t> &bounds() const { return p_; }
    // all-gather schedule
    std::size_t chunks() const { return st_->chunk_bounds.size() - 1; }
    const std::vector<v_t> &chunk_bounds() const { return st_->chunk_bounds; }
    const matrix_t &remote_chunk(std::size_t i, std::size_t c) const { need_chunks(); return st_->chunks[i][c]; }
    // halo schedule
    const matrix_t &halo_remote(std::size_t i) const { need_halo(); return st_->halo_remote[i]; }
    const std::vector<v_t> &halo_need(std::size_t i, std::size_t s) const { need_halo(); return st_->need[i][s]; }
    // rows moved per (receiver, source) pair: the matrix test/data/prep.py:237-244 prints for a partition
    std::vector<std::vector<std::size_t>> halo_volume() const {
        need_halo();
        std::vector<std::vector<std::size_t>> V(size(), std::vector<std::size_t>(size(), 0));
        for (std::size_t i = 0; i < size(); i++)
            for (std::size_t s = 0; s < size(); s++) V[i][s] = st_->need[i][s].size();
        return V;
    }
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
        auto recv = std::make_shared<std::vector<float *>>();
        for (std::size_t j = 0; j < size(); j++) recv->push_back(bAs[j].buffer());
        const float *root = As[i].buffer();
        const std::size_t count = As[i].size();
        if (!ctx.threaded()) {
            const auto streams = ctx.streams(stream_id);
            mggcn_comm_broadcast_f32(ctx.comm(), root, recv->data(), count, (int)i, streams.data());
            return;
        }
        for (std::size_t j = 0; j < size(); j++)
            ctx.on(j, [cm = ctx.comm(), j, root, recv, count, i, st = ctx[j].stream(stream_id)] {
                mggcn_comm_broadcast_rank_f32(cm, (int)j, root, recv->data(), count, (int)i, st);
            });
    }

    // rows [row_begin, row_end) of EVERY shard to every GPU, rank order, at gathered[j] + P * row_begin * m:
    // one piece of the MI355X-first exchange (the whole shard when the range is the whole shard)
    void allgather(const dist_context &ctx, const std::vector<matrix_t> &gathered, std::size_t row_begin, std::size_t row_end,
                   int stream_id = 1) const {
        auto send = std::make_shared<std::vector<const float *>>();
        auto recv = std::make_shared<std::vector<float *>>();
        const std::size_t P = size(), w = m();
        for (std::size_t j = 0; j < P; j++) {
            send->push_back(As[j].buffer() + row_begin * w);
            recv->push_back(gathered[j].buffer() + P * row_begin * w);
        }
        const std::size_t count = (row_end - row_begin) * w;
        if (!ctx.threaded()) {
            const auto streams = ctx.streams(stream_id);
            mggcn_comm_allgather_f32(ctx.comm(), send->data(), recv->data(), count, streams.data());
            return;
        }
        for (std::size_t j = 0; j < P; j++)
            ctx.on(j, [cm = ctx.comm(), j, send, recv, count, st = ctx[j].stream(stream_id)] {
                mggcn_comm_allgather_rank_f32(cm, (int)j, send->data(), recv->data(), count, st);
            });
    }
    void allgather(const dist_context &ctx, const std::vector<matrix_t> &gathered, int stream_id = 1) const {
        allgather(ctx, gathered, 0, As[0].n(), stream_id);
    }

    void zero(const dist_context &ctx) const { for (std::size_t i = 0; i < size(); i++) ctx.on(i, [c = ctx[i], a = As[i]] { a.zero(c); }); }

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
    // views of caller-owned per-GPU buffers (G_W and G_b of a layer share one: a single all-reduce)
    repl_dn_matrix(const dist_context &ctx, std::size_t N, std::size_t M, const std::vector<mggcn::device_ptr<r_t>> &buffers) {
        for (std::size_t i = 0; i < ctx.size(); i++) { ctx[i].set(); As.emplace_back(N, M, buffers[i]); }
    }

    auto n() const { return As[0].n(); }
    auto m() const { return As[0].m(); }
    auto shape() const { return std::make_pair(n(), m()); }
    auto size() const { return As.size(); }
    const matrix_t &operator[](std::size_t i) const { return As[i]; }

    // in-place sum over the GPUs (reference :587-592), on stream `stream_id` of every GPU
    void allreduce(const dist_context &ctx, int stream_id = 0) const {
        auto bufs = std::make_shared<std::vector<float *>>();
        for (const auto &A : As) bufs->push_back(A.buffer());
        const std::size_t count = As[0].size();
        if (!ctx.threaded()) {
            const auto streams = ctx.streams(stream_id);
            mggcn_comm_allreduce_sum_f32(ctx.comm(), bufs->data(), count, streams.data());
            return;
        }
        for (std::size_t j = 0; j < As.size(); j++)
            ctx.on(j, [cm = ctx.comm(), j, bufs, count, st = ctx[j].stream(stream_id)] {
                mggcn_comm_allreduce_sum_rank_f32(cm, (int)j, bufs->data(), count, st);
            });
    }

    // the reference initialises GPU 0 and broadcasts (:601-609); the seed-99 host generator
    // gives every GPU the same bits without traffic
    void init(const dist_context &ctx, r_t gain = (r_t)std::sqrt(2 / (1 + 0.01 * 0.01))) {
        for (std::size_t i = 0; i < As.size(); i++) { ctx[i].set(); As[i].init(gain); }
    }
    void fill(const dist_context &ctx, r_t v) {
        for (std::size_t i = 0; i < As.size(); i++) { ctx[i].set(); As[i].fill(v); }
    }
    void zero(const dist_context &ctx) const { for (std::size_t i = 0; i < As.size(); i++) ctx.on(i, [c = ctx[i], a = As[i]] { a.zero(c); }); }
};
