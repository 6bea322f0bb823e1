// comm.cpp -- libmggcn_comm.so: the collectives of the single-process, P-GPU host layer.
// See include/mggcn_comm.h for the reference call sites this replaces.
//
// Two transports behind the same entry points:
//   rccl  one RCCL communicator per GPU (ncclCommInitAll).  The all-ranks entry points issue a
//         group of P per-communicator calls from the calling thread -- the reference's NCCL
//         pattern (src/dist_matrix.hpp:26-31, :458-467, :587-592); the per-rank entry points
//         issue rank j's call from rank j's enqueue thread (one thread per device needs no
//         group).  Default when the P ranks sit on P different GPUs.
//   p2p   device-to-device copies (hipMemcpyPeerAsync over xGMI: copy engines, no compute
//         unit is taken from the SpMM that runs meanwhile; same-device copies when ranks share
//         a GPU) ordered by events, PER PAIR: receiver j pulls from sender i as soon as i's
//         stream has produced the piece (`ready`), every pull from another GPU on its own
//         per-peer stream so that the seven links of a GPU work at once (ranks that share a
//         device pull on the caller's stream); nobody waits for a third rank.  A sender
//         may overwrite what it sent once its readers are done: that wait is either issued
//         right after the exchange (default: NCCL's contract) or deferred to an explicit
//         mggcn_comm_release (MGGCN_COMM_DEFER_RELEASE: the host layer releases on the
//         compute stream at the end of an SpMM, so a late rank delays nobody's SpMM over
//         pieces that have landed).  Chosen automatically when two ranks share a GPU (RCCL
//         refuses that) -- which is how the P > 1 schedules run on a one-GPU box -- or with
//         MGGCN_COMM_TRANSPORT=p2p.  Sums are formed in rank order on every GPU: identical
//         bits everywhere.
//
// Host-side protocol of the p2p transport (all ranks take part in every exchange, in the same
// order; exchange number s = 1, 2, ...):
//   begin(j)    record ready[j][s % R] on j's stream, publish ready_seq[j] = s
//   pull(j)     for every sender i it reads: wait (host) until ready_seq[i] >= s, then make the
//               pulling stream wait for ready[i][s % R]; copies; record done[j][s % R] on j's
//               stream, publish done_seq[j] = s
//   release(j)  wait (host) until done_seq[i] >= s_last for every other rank, make the given
//               stream wait for done[i][s_last % R] (pulls of one rank run in order, so the
//               latest `done` covers the earlier ones), publish released_seq[j]
// The all-ranks entry points run begin for all ranks, then pull for all, then release: the host
// waits are satisfied by construction.  The per-rank entry points run on P threads and the
// sequence counters are what orders "event recorded" before "event waited for".  Event slots
// are reused every R exchanges: a rank that is R / 2 exchanges behind with its releases
// releases on the spot, so a slot is never re-recorded while somebody still has to wait for it.
//
// MGGCN_P2P_PUSH=1 turns the copies round: the SENDER writes into the receiver's buffer (posted
// writes over xGMI instead of read round trips; the copy runs on the sender's copy engines):
//   begin(j)    as above: ready[j][s % R] now also says "my receive buffer is free" (everything
//               that read it was queued on j's stream before)
//   push(j)     for every receiver i of a piece of mine: wait (host) until ready_seq[i] >= s, make
//               the pushing stream (one per peer on another GPU) wait for ready[i][s % R], copy,
//               record pushed[j][i][s % R]; j's stream joins its pushing streams and records
//               done[j][s % R]; publish pushed_seq[j] = s
//   receive(j)  for every sender i of a piece of mine: wait (host) until pushed_seq[i] >= s, make
//               j's stream wait for pushed[i][j][s % R]; publish received_seq[j] = s
//   release(j)  local: the given stream waits for my own done[j][s_last % R] -- what a sender may
//               overwrite depends on its own copies only, no rank waits for another's progress
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <set>
#include <thread>
#include <vector>

#include "comm_layout.h"
#include "mggcn_comm.h"

#define MGGCN_API extern "C" __attribute__((visibility("default")))

#define CHECK_RCCL(expr)                                                                         \
    do {                                                                                         \
        ncclResult_t r_ = (expr);                                                                \
        if (r_ != ncclSuccess) {                                                                 \
            std::fprintf(stderr, "MGGCN RCCL failed at %s:%d '%s'\n", __FILE__, __LINE__,        \
                         ncclGetErrorString(r_));                                                \
            std::exit(EXIT_FAILURE);                                                             \
        }                                                                                        \
    } while (0)

#define CHECK_HIP(expr)                                                                          \
    do {                                                                                         \
        hipError_t s_ = (expr);                                                                  \
        if (s_ != hipSuccess) {                                                                  \
            std::fprintf(stderr, "MGGCN HIP API failed at %s:%d with error: %s (%d)\n", __FILE__, \
                         __LINE__, hipGetErrorString(s_), (int)s_);                              \
            std::exit(EXIT_FAILURE);                                                             \
        }                                                                                        \
    } while (0)

namespace {

constexpr int kRing = 32;                      // event slots per rank (see the header comment)

struct rank_state {
    hipEvent_t ready[kRing] = {}, done[kRing] = {};
    hipEvent_t fork = nullptr;                 // j's stream -> its per-peer streams
    std::vector<hipStream_t> peer_stream;      // [i]: pulls from rank i (null: pull on j's own stream)
    std::vector<hipEvent_t> peer_done;         // [i]
    std::vector<hipEvent_t> pushed;            // push form: [i * kRing + slot]: my pieces for rank i have been written
    std::atomic<std::uint64_t> ready_seq{0}, done_seq{0}, released_seq{0};
    std::atomic<std::uint64_t> pushed_seq{0}, received_seq{0};          // push form
    std::uint64_t seq = 0;                     // exchanges begun; touched by the rank's caller only
    float *scratch = nullptr;                  // all-reduce: P x count floats
    std::size_t scratch_floats = 0;
};

}  // namespace

struct mggcn_comm {
    std::vector<ncclComm_t> comms;      // rccl transport; empty for p2p
    std::vector<int> devices;
    bool p2p = false;
    bool push = false;                  // p2p: senders write (MGGCN_P2P_PUSH=1) instead of receivers reading
    unsigned flags = 0;                 // MGGCN_COMM_DEFER_RELEASE | MGGCN_COMM_SKIP_SELF
    std::vector<std::unique_ptr<rank_state>> rk;   // p2p
};

namespace {

inline hipStream_t as_stream(mggcn_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
inline int size_of(const mggcn_comm *c) { return (int)c->devices.size(); }

void require(bool ok, const char *what) {
    if (ok) return;
    std::fprintf(stderr, "MGGCN precondition failed: %s\n", what);
    std::exit(EXIT_FAILURE);
}

// host wait for another rank's progress; a peer that never arrives is a protocol error (or a dead enqueue thread):
// say so instead of hanging
void await(const std::atomic<std::uint64_t> &counter, std::uint64_t value, const char *what, int me, int peer) {
    if (counter.load(std::memory_order_acquire) >= value) return;
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0;; spins++) {
        if (counter.load(std::memory_order_acquire) >= value) return;
        if (spins < 4096) continue;
        std::this_thread::yield();
        if ((spins & 0xffffu) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) {
            std::fprintf(stderr, "MGGCN comm: rank %d waited 120 s for rank %d to reach %s of exchange %llu (it is at %llu)\n", me, peer,
                         what, (unsigned long long)value, (unsigned long long)counter.load());
            std::exit(EXIT_FAILURE);
        }
    }
}

void copy_f32(const mggcn_comm *c, float *dst, int dst_rank, const float *src, int src_rank, size_t count, hipStream_t st) {
    if (!count) return;
    const int dd = c->devices[dst_rank], sd = c->devices[src_rank];
    if (dd == sd) CHECK_HIP(hipMemcpyAsync(dst, src, count * sizeof(float), hipMemcpyDeviceToDevice, st));
    else CHECK_HIP(hipMemcpyPeerAsync(dst, dd, src, sd, count * sizeof(float), st));
}

void p2p_release_rank(mggcn_comm *c, int j, hipStream_t st) {
    rank_state &me = *c->rk[j];
    const std::uint64_t s = me.seq;
    if (me.released_seq.load(std::memory_order_relaxed) >= s) return;
    if (c->push) {                                                 // my send buffers depend on my own copies only
        if (s) CHECK_HIP(hipStreamWaitEvent(st, me.done[s % kRing], 0));
        me.released_seq.store(s, std::memory_order_release);
        return;
    }
    for (int i = 0; i < size_of(c); i++) {
        if (i == j) continue;
        await(c->rk[i]->done_seq, s, "the end", j, i);
        CHECK_HIP(hipStreamWaitEvent(st, c->rk[i]->done[s % kRing], 0));
    }
    me.released_seq.store(s, std::memory_order_release);
}

// starts exchange number (returned) for rank j: everything queued on `st` so far is what the peers may read
std::uint64_t p2p_begin(mggcn_comm *c, int j, hipStream_t st) {
    rank_state &me = *c->rk[j];
    CHECK_HIP(hipSetDevice(c->devices[j]));
    if (me.seq - me.released_seq.load(std::memory_order_relaxed) >= (std::uint64_t)kRing / 2) p2p_release_rank(c, j, st);
    const std::uint64_t s = ++me.seq;
    if (s > (std::uint64_t)kRing)                                  // slot s % R was last used by exchange s - R
        for (int i = 0; i < size_of(c); i++) {
            if (i == j) continue;
            if (c->push) {
                await(c->rk[i]->pushed_seq, s - kRing, "the pushes", j, i);         // i has waited for my ready[s - R] ...
                await(c->rk[i]->received_seq, s - kRing, "the receipt", j, i);      // ... and for my pushed[i][s - R]
            } else {
                await(c->rk[i]->done_seq, s - kRing, "the end", j, i);              // i has waited for my ready[s - R] ...
                await(c->rk[i]->released_seq, s - kRing, "the release", j, i);      // ... and for my done[s - R]
            }
        }
    CHECK_HIP(hipEventRecord(me.ready[s % kRing], st));
    me.ready_seq.store(s, std::memory_order_release);
    return s;
}

struct pull_item { int src; float *dst; const float *from; size_t count; };

// rank j's copies of exchange s: each from its sender as soon as that sender is ready
void p2p_pull(mggcn_comm *c, int j, std::uint64_t s, hipStream_t st, const std::vector<pull_item> &items) {
    rank_state &me = *c->rk[j];
    CHECK_HIP(hipSetDevice(c->devices[j]));
    bool forked = false;
    std::vector<char> joined(size_of(c), 0);
    for (const auto &it : items) {
        if (!it.count) continue;
        if (it.src == j) {                                         // my own piece: nobody to wait for
            copy_f32(c, it.dst, j, it.from, j, it.count, st);
            continue;
        }
        await(c->rk[it.src]->ready_seq, s, "the start", j, it.src);
        hipEvent_t ready = c->rk[it.src]->ready[s % kRing];
        hipStream_t ps = me.peer_stream[it.src];
        if (!ps) {                                                 // single-stream form: pulls one after the other
            CHECK_HIP(hipStreamWaitEvent(st, ready, 0));
            copy_f32(c, it.dst, j, it.from, it.src, it.count, st);
            continue;
        }
        if (!forked) { CHECK_HIP(hipEventRecord(me.fork, st)); forked = true; }
        CHECK_HIP(hipStreamWaitEvent(ps, me.fork, 0));             // the receive buffer is free (st has seen its last readers)
        CHECK_HIP(hipStreamWaitEvent(ps, ready, 0));
        copy_f32(c, it.dst, j, it.from, it.src, it.count, ps);
        joined[it.src] = 1;
    }
    for (int i = 0; i < size_of(c); i++)
        if (joined[i]) {
            CHECK_HIP(hipEventRecord(me.peer_done[i], me.peer_stream[i]));
            CHECK_HIP(hipStreamWaitEvent(st, me.peer_done[i], 0));
        }
    CHECK_HIP(hipEventRecord(me.done[s % kRing], st));
    me.done_seq.store(s, std::memory_order_release);
}

// push form, rank j's copies of exchange s: my pieces into every receiver's buffer as soon as that buffer is free.
// all_items[i] = what receiver i gets (the pull form's list); mine are the entries with src == j.
void p2p_push(mggcn_comm *c, int j, std::uint64_t s, hipStream_t st, const std::vector<std::vector<pull_item>> &all_items) {
    rank_state &me = *c->rk[j];
    CHECK_HIP(hipSetDevice(c->devices[j]));
    for (const auto &it : all_items[j])                            // my own piece: nobody to wait for
        if (it.src == j && it.count) copy_f32(c, it.dst, j, it.from, j, it.count, st);
    bool forked = false;
    std::vector<char> joined(size_of(c), 0);
    for (int i = 0; i < size_of(c); i++) {
        if (i == j) continue;
        bool any = false;
        hipStream_t ps = me.peer_stream[i];
        for (const auto &it : all_items[i]) {
            if (it.src != j || !it.count) continue;
            if (!any) {
                await(c->rk[i]->ready_seq, s, "the start", j, i);
                hipEvent_t free_i = c->rk[i]->ready[s % kRing];   // i's receive buffer is free
                if (ps) {
                    if (!forked) { CHECK_HIP(hipEventRecord(me.fork, st)); forked = true; }
                    CHECK_HIP(hipStreamWaitEvent(ps, me.fork, 0)); // my piece has been produced
                    CHECK_HIP(hipStreamWaitEvent(ps, free_i, 0));
                } else {
                    CHECK_HIP(hipStreamWaitEvent(st, free_i, 0));
                }
                any = true;
            }
            copy_f32(c, it.dst, i, it.from, j, it.count, ps ? ps : st);
        }
        if (!any) continue;
        CHECK_HIP(hipEventRecord(me.pushed[(size_t)i * kRing + s % kRing], ps ? ps : st));
        if (ps) joined[i] = 1;
    }
    for (int i = 0; i < size_of(c); i++)
        if (joined[i]) CHECK_HIP(hipStreamWaitEvent(st, me.pushed[(size_t)i * kRing + s % kRing], 0));
    CHECK_HIP(hipEventRecord(me.done[s % kRing], st));             // my send buffers are mine again after this
    me.done_seq.store(s, std::memory_order_release);
    me.pushed_seq.store(s, std::memory_order_release);
}

// push form: rank j's stream waits for the pieces the others wrote into its buffer
void p2p_receive(mggcn_comm *c, int j, std::uint64_t s, hipStream_t st, const std::vector<pull_item> &items) {
    rank_state &me = *c->rk[j];
    CHECK_HIP(hipSetDevice(c->devices[j]));
    std::vector<char> from(size_of(c), 0);
    for (const auto &it : items)
        if (it.count && it.src != j) from[it.src] = 1;
    for (int i = 0; i < size_of(c); i++) {
        if (!from[i]) continue;
        await(c->rk[i]->pushed_seq, s, "the pushes", j, i);
        CHECK_HIP(hipStreamWaitEvent(st, c->rk[i]->pushed[(size_t)j * kRing + s % kRing], 0));
    }
    me.received_seq.store(s, std::memory_order_release);
}

void p2p_finish(mggcn_comm *c, int j, hipStream_t st, bool may_defer) {
    if (may_defer && (c->flags & MGGCN_COMM_DEFER_RELEASE)) return;
    p2p_release_rank(c, j, st);
}

// ---- what each rank pulls, per collective ---------------------------------------------------
std::vector<pull_item> broadcast_items(const mggcn_comm *, int j, const float *send_root, float *const *recv, size_t count, int root) {
    if (recv[j] == send_root) return {};
    return {{root, recv[j], send_root, count}};
}

std::vector<pull_item> allgather_items(const mggcn_comm *c, int j, const float *const *send, float *const *recv, size_t count) {
    std::vector<pull_item> v;
    for (int i = 0; i < size_of(c); i++) {
        float *dst = recv[j] + (size_t)i * count;
        if (dst == send[i]) continue;
        if (i == j && (c->flags & MGGCN_COMM_SKIP_SELF)) continue;
        v.push_back({i, dst, send[i], count});
    }
    return v;
}

std::vector<pull_item> alltoallv_items(const mggcn_comm *c, int k, const float *const *send, float *const *recv,
                                       const size_t *counts, const std::vector<size_t> &sdis, const std::vector<size_t> &rdis) {
    const int P = size_of(c);
    std::vector<pull_item> v;
    for (int j = 0; j < P; j++)
        v.push_back({j, recv[k] + rdis[(size_t)k * P + j], send[j] + sdis[(size_t)j * P + k], counts[(size_t)j * P + k]});
    return v;
}

void allreduce_scratch(mggcn_comm *c, int j, size_t count) {
    rank_state &me = *c->rk[j];
    const size_t need = (size_t)size_of(c) * count;
    if (me.scratch_floats >= need) return;
    CHECK_HIP(hipSetDevice(c->devices[j]));
    CHECK_HIP(hipDeviceSynchronize());
    if (me.scratch) CHECK_HIP(hipFree(me.scratch));
    CHECK_HIP(hipMalloc(&me.scratch, need * sizeof(float)));
    me.scratch_floats = need;
}

std::vector<pull_item> allreduce_items(mggcn_comm *c, int j, float *const *bufs, size_t count) {
    std::vector<pull_item> v;
    for (int i = 0; i < size_of(c); i++) v.push_back({i, c->rk[j]->scratch + (size_t)i * count, bufs[i], count});
    return v;
}

// every rank holds all P contributions and nobody reads bufs[] any more: sum in rank order
void allreduce_sum_local(mggcn_comm *c, int j, float *buf, size_t count, mggcn_stream_t stream) {
    CHECK_HIP(hipSetDevice(c->devices[j]));
    CHECK_HIP(hipMemcpyAsync(buf, c->rk[j]->scratch, count * sizeof(float), hipMemcpyDeviceToDevice, as_stream(stream)));
    for (int i = 1; i < size_of(c); i++) mggcn_axpy_f32(stream, c->rk[j]->scratch + (size_t)i * count, buf, 1.0f, count);
}

}  // namespace

MGGCN_API mggcn_comm *mggcn_comm_init_all(int P, const int *devices) {
    require(P > 0, "communicator size must be positive");
    auto *c = new mggcn_comm;
    c->devices.resize(P);
    for (int i = 0; i < P; i++) c->devices[i] = devices ? devices[i] : i;
    const std::set<int> distinct(c->devices.begin(), c->devices.end());
    const char *tr = std::getenv("MGGCN_COMM_TRANSPORT");
    c->p2p = (int)distinct.size() != P || (tr && std::strcmp(tr, "p2p") == 0);
    require(!(tr && std::strcmp(tr, "rccl") == 0 && (int)distinct.size() != P), "MGGCN_COMM_TRANSPORT=rccl needs one GPU per rank");
    if (!c->p2p) {
        c->comms.resize(P);
        CHECK_RCCL(ncclCommInitAll(c->comms.data(), P, c->devices.data()));
        return c;
    }
    for (int a : distinct)
        for (int b : distinct) {
            if (a == b) continue;
            int can = 0;
            CHECK_HIP(hipDeviceCanAccessPeer(&can, a, b));
            if (!can) continue;                        // hipMemcpyPeerAsync stages through the host then
            CHECK_HIP(hipSetDevice(a));
            const hipError_t e = hipDeviceEnablePeerAccess(b, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) CHECK_HIP(e);
            (void)hipGetLastError();
        }
    // One pulling stream per (receiver, sender) pair ON DIFFERENT GPUs, so that a GPU's links work side by side.  Ranks that
    // share a device pull from each other on the caller's stream, one copy after the other: there is no link to keep busy, and
    // eight ranks x seven extra streams on ONE card doubled the epoch of `mg_gcn -P 8` there (105.9 against 52.1 ms: the runtime
    // multiplexes the streams over four hardware queues; profiles/experiments/cli_p8_variants_r04.log).
    // MGGCN_P2P_PEER_STREAMS=0: never; =1: always, same device or not (tests: the only way to run this path on a one-GPU box).
    const char *push = std::getenv("MGGCN_P2P_PUSH");
    c->push = push && std::atoi(push) != 0;
    const char *ps = std::getenv("MGGCN_P2P_PEER_STREAMS");
    const int peer_mode = ps ? std::atoi(ps) : -1;                 // -1: across devices only
    for (int j = 0; j < P; j++) {
        c->rk.push_back(std::make_unique<rank_state>());
        rank_state &r = *c->rk.back();
        CHECK_HIP(hipSetDevice(c->devices[j]));
        for (int k = 0; k < kRing; k++) {
            CHECK_HIP(hipEventCreateWithFlags(&r.ready[k], hipEventDisableTiming));
            CHECK_HIP(hipEventCreateWithFlags(&r.done[k], hipEventDisableTiming));
        }
        CHECK_HIP(hipEventCreateWithFlags(&r.fork, hipEventDisableTiming));
        if (c->push) {
            r.pushed.assign((size_t)P * kRing, nullptr);
            for (auto &e : r.pushed) CHECK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        }
        r.peer_stream.assign(P, nullptr);
        r.peer_done.assign(P, nullptr);
        int least = 0, greatest = 0;
        CHECK_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
        for (int i = 0; i < P && peer_mode != 0 && P > 2; i++) {
            if (i == j || (peer_mode < 0 && c->devices[i] == c->devices[j])) continue;
            CHECK_HIP(hipStreamCreateWithPriority(&r.peer_stream[i], hipStreamNonBlocking, greatest));
            CHECK_HIP(hipEventCreateWithFlags(&r.peer_done[i], hipEventDisableTiming));
        }
    }
    return c;
}

MGGCN_API void mggcn_comm_destroy(mggcn_comm *comm) {
    if (!comm) return;
    for (auto &c : comm->comms) ncclCommDestroy(c);
    for (size_t j = 0; j < comm->rk.size(); j++) {
        rank_state &r = *comm->rk[j];
        (void)hipSetDevice(comm->devices[j]);
        for (int k = 0; k < kRing; k++) { (void)hipEventDestroy(r.ready[k]); (void)hipEventDestroy(r.done[k]); }
        (void)hipEventDestroy(r.fork);
        for (auto &s : r.peer_stream) if (s) (void)hipStreamDestroy(s);
        for (auto &e : r.peer_done) if (e) (void)hipEventDestroy(e);
        for (auto &e : r.pushed) if (e) (void)hipEventDestroy(e);
        if (r.scratch) (void)hipFree(r.scratch);
    }
    delete comm;
}

MGGCN_API int mggcn_comm_size(const mggcn_comm *comm) { return size_of(comm); }

MGGCN_API const char *mggcn_comm_transport(const mggcn_comm *comm) { return !comm->p2p ? "rccl" : comm->push ? "p2p-push" : "p2p"; }

MGGCN_API void mggcn_comm_set_exchange_flags(mggcn_comm *comm, unsigned flags) { comm->flags = flags; }

MGGCN_API void mggcn_comm_release(mggcn_comm *comm, const mggcn_stream_t *streams) {
    if (!comm->p2p) return;
    for (int j = 0; j < size_of(comm); j++) {
        CHECK_HIP(hipSetDevice(comm->devices[j]));
        p2p_release_rank(comm, j, as_stream(streams[j]));
    }
}

MGGCN_API void mggcn_comm_release_rank(mggcn_comm *comm, int rank, mggcn_stream_t stream) {
    if (!comm->p2p) return;
    CHECK_HIP(hipSetDevice(comm->devices[rank]));
    p2p_release_rank(comm, rank, as_stream(stream));
}

// ---- all ranks from the calling thread --------------------------------------------------------
namespace {
template <typename Items>
void p2p_all_ranks(mggcn_comm *c, const mggcn_stream_t *streams, bool may_defer, Items &&items) {
    const int P = size_of(c);
    std::vector<std::uint64_t> s(P);
    for (int j = 0; j < P; j++) s[j] = p2p_begin(c, j, as_stream(streams[j]));
    if (c->push) {
        std::vector<std::vector<pull_item>> all(P);
        for (int j = 0; j < P; j++) all[j] = items(j);
        for (int j = 0; j < P; j++) p2p_push(c, j, s[j], as_stream(streams[j]), all);
        for (int j = 0; j < P; j++) p2p_receive(c, j, s[j], as_stream(streams[j]), all[j]);
    } else
        for (int j = 0; j < P; j++) p2p_pull(c, j, s[j], as_stream(streams[j]), items(j));
    for (int j = 0; j < P; j++) {
        CHECK_HIP(hipSetDevice(c->devices[j]));
        p2p_finish(c, j, as_stream(streams[j]), may_defer);
    }
}
}  // namespace

MGGCN_API void mggcn_comm_broadcast_f32(mggcn_comm *comm, const float *send_root, float *const *recv, size_t count,
                                        int root, const mggcn_stream_t *streams) {
    const int P = size_of(comm);
    if (comm->p2p) {
        p2p_all_ranks(comm, streams, true, [&](int j) { return broadcast_items(comm, j, send_root, recv, count, root); });
        return;
    }
    CHECK_RCCL(ncclGroupStart());
    for (int j = 0; j < P; j++) {
        CHECK_HIP(hipSetDevice(comm->devices[j]));
        CHECK_RCCL(ncclBroadcast(send_root, recv[j], count, ncclFloat32, root, comm->comms[j], as_stream(streams[j])));
    }
    CHECK_RCCL(ncclGroupEnd());
}

MGGCN_API void mggcn_comm_allgather_f32(mggcn_comm *comm, const float *const *send, float *const *recv, size_t count,
                                        const mggcn_stream_t *streams) {
    const int P = size_of(comm);
    if (comm->p2p) {
        p2p_all_ranks(comm, streams, true, [&](int j) { return allgather_items(comm, j, send, recv, count); });
        return;
    }
    CHECK_RCCL(ncclGroupStart());
    for (int j = 0; j < P; j++) {
        CHECK_HIP(hipSetDevice(comm->devices[j]));
        CHECK_RCCL(ncclAllGather(send[j], recv[j], count, ncclFloat32, comm->comms[j], as_stream(streams[j])));
    }
    CHECK_RCCL(ncclGroupEnd());
}

MGGCN_API void mggcn_comm_alltoallv_displacements(int P, const size_t *counts, size_t *sdis, size_t *rdis) {
    mggcn_layout::alltoallv_displacements(P, counts, sdis, rdis);
}

namespace {
void rccl_alltoallv_rank(mggcn_comm *comm, int j, const float *const *send, float *const *recv, const size_t *counts,
                         const std::vector<size_t> &sdis, const std::vector<size_t> &rdis, mggcn_stream_t stream) {
    const int P = size_of(comm);
    CHECK_HIP(hipSetDevice(comm->devices[j]));
    for (int k = 0; k < P; k++) {
        const size_t out = counts[(size_t)j * P + k], in = counts[(size_t)k * P + j];
        if (out) CHECK_RCCL(ncclSend(send[j] + sdis[(size_t)j * P + k], out, ncclFloat32, k, comm->comms[j], as_stream(stream)));
        if (in) CHECK_RCCL(ncclRecv(recv[j] + rdis[(size_t)j * P + k], in, ncclFloat32, k, comm->comms[j], as_stream(stream)));
    }
}
}  // namespace

MGGCN_API void mggcn_comm_alltoallv_f32(mggcn_comm *comm, const float *const *send, float *const *recv,
                                        const size_t *counts, const mggcn_stream_t *streams) {
    const int P = size_of(comm);
    std::vector<size_t> sdis((size_t)P * P, 0), rdis((size_t)P * P, 0);
    mggcn_comm_alltoallv_displacements(P, counts, sdis.data(), rdis.data());
    if (comm->p2p) {
        p2p_all_ranks(comm, streams, true, [&](int k) { return alltoallv_items(comm, k, send, recv, counts, sdis, rdis); });
        return;
    }
    CHECK_RCCL(ncclGroupStart());
    for (int j = 0; j < P; j++) rccl_alltoallv_rank(comm, j, send, recv, counts, sdis, rdis, streams[j]);
    CHECK_RCCL(ncclGroupEnd());
}

MGGCN_API void mggcn_comm_allreduce_sum_f32(mggcn_comm *comm, float *const *bufs, size_t count,
                                            const mggcn_stream_t *streams) {
    const int P = size_of(comm);
    if (comm->p2p) {
        if (P == 1 || count == 0) return;
        for (int j = 0; j < P; j++) allreduce_scratch(comm, j, count);
        // never deferred: bufs[] is overwritten with the sum right away
        p2p_all_ranks(comm, streams, false, [&](int j) { return allreduce_items(comm, j, bufs, count); });
        for (int j = 0; j < P; j++) allreduce_sum_local(comm, j, bufs[j], count, streams[j]);
        return;
    }
    CHECK_RCCL(ncclGroupStart());
    for (int j = 0; j < P; j++) {
        CHECK_HIP(hipSetDevice(comm->devices[j]));
        CHECK_RCCL(ncclAllReduce(bufs[j], bufs[j], count, ncclFloat32, ncclSum, comm->comms[j], as_stream(streams[j])));
    }
    CHECK_RCCL(ncclGroupEnd());
}

// ---- one rank's share, from that rank's enqueue thread ----------------------------------------
namespace {
// peer_state: the lists name memory the RECEIVER allocates before it begins (the all-reduce scratch): read them only
// after every receiver has begun this exchange
template <typename Items>
void p2p_one_rank(mggcn_comm *c, int j, mggcn_stream_t stream, bool may_defer, Items &&items, bool peer_state = false) {
    const std::uint64_t s = p2p_begin(c, j, as_stream(stream));
    if (c->push) {
        std::vector<std::vector<pull_item>> all(size_of(c));
        for (int i = 0; i < size_of(c) && peer_state; i++)
            if (i != j) await(c->rk[i]->ready_seq, s, "the start", j, i);
        for (int i = 0; i < size_of(c); i++) all[i] = items(i);
        p2p_push(c, j, s, as_stream(stream), all);
        p2p_receive(c, j, s, as_stream(stream), all[j]);
    } else
        p2p_pull(c, j, s, as_stream(stream), items(j));
    p2p_finish(c, j, as_stream(stream), may_defer);
}
}  // namespace

MGGCN_API void mggcn_comm_broadcast_rank_f32(mggcn_comm *comm, int rank, const float *send_root, float *const *recv,
                                             size_t count, int root, mggcn_stream_t stream) {
    if (comm->p2p) {
        p2p_one_rank(comm, rank, stream, true, [&](int i) { return broadcast_items(comm, i, send_root, recv, count, root); });
        return;
    }
    CHECK_HIP(hipSetDevice(comm->devices[rank]));
    CHECK_RCCL(ncclBroadcast(send_root, recv[rank], count, ncclFloat32, root, comm->comms[rank], as_stream(stream)));
}

MGGCN_API void mggcn_comm_allgather_rank_f32(mggcn_comm *comm, int rank, const float *const *send, float *const *recv,
                                             size_t count, mggcn_stream_t stream) {
    if (comm->p2p) {
        p2p_one_rank(comm, rank, stream, true, [&](int i) { return allgather_items(comm, i, send, recv, count); });
        return;
    }
    CHECK_HIP(hipSetDevice(comm->devices[rank]));
    CHECK_RCCL(ncclAllGather(send[rank], recv[rank], count, ncclFloat32, comm->comms[rank], as_stream(stream)));
}

MGGCN_API void mggcn_comm_alltoallv_rank_f32(mggcn_comm *comm, int rank, const float *const *send, float *const *recv,
                                             const size_t *counts, mggcn_stream_t stream) {
    const int P = size_of(comm);
    std::vector<size_t> sdis((size_t)P * P, 0), rdis((size_t)P * P, 0);
    mggcn_comm_alltoallv_displacements(P, counts, sdis.data(), rdis.data());
    if (comm->p2p) {
        p2p_one_rank(comm, rank, stream, true, [&](int i) { return alltoallv_items(comm, i, send, recv, counts, sdis, rdis); });
        return;
    }
    CHECK_RCCL(ncclGroupStart());                     // this rank's sends and receives are one operation
    rccl_alltoallv_rank(comm, rank, send, recv, counts, sdis, rdis, stream);
    CHECK_RCCL(ncclGroupEnd());
}

MGGCN_API void mggcn_comm_allreduce_sum_rank_f32(mggcn_comm *comm, int rank, float *const *bufs, size_t count,
                                                 mggcn_stream_t stream) {
    const int P = size_of(comm);
    if (comm->p2p) {
        if (P == 1 || count == 0) return;
        allreduce_scratch(comm, rank, count);
        p2p_one_rank(comm, rank, stream, false, [&](int i) { return allreduce_items(comm, i, bufs, count); }, true);
        allreduce_sum_local(comm, rank, bufs[rank], count, stream);
        return;
    }
    CHECK_HIP(hipSetDevice(comm->devices[rank]));
    CHECK_RCCL(ncclAllReduce(bufs[rank], bufs[rank], count, ncclFloat32, ncclSum, comm->comms[rank], as_stream(stream)));
}
