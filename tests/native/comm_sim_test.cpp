// tests/native/comm_sim_test.cpp -- the peer-copy transport of libmggcn_comm.so (mg-gcn_amd/csrc/comm.cpp, compiled as is)
// against a MODEL of HIP's stream / event semantics (tests/native/hipsim/), on the CPU, under the sanitizers.
//
// Why: the transport's cross-device form -- one pulling / pushing stream per peer GPU, events recorded on one device and
// waited for on another, host sequence counters that order "recorded" before "waited for" across P enqueue threads, event
// slots re-used every 32 exchanges -- takes that form only between DIFFERENT GPUs, and the builder's pool has one-GPU boxes.
// Here every rank is its own "device", nothing runs when it is enqueued, and a drain executes the queued operations in a
// random or adversarial order that respects nothing but stream order and the event waits the transport asked for.
//
// The RCCL transport runs too, on a model of RCCL's contract (hipsim/rccl/rccl.h): its multi-rank branches -- grouped calls from one
// thread, per-rank calls from P threads, the send / receive all-to-all -- have never met more than one real rank either.
// Each scenario: P ranks x {receivers pull, senders push, RCCL} x {P enqueue threads + per-rank entry points, one thread + all-ranks
// entry points} x exchange flags, several iterations of all-gather / broadcast / all-to-all / all-reduce.  Around every
// exchange a rank FILLS what it sends (a kernel on its stream, values that name rank / iteration / position), CHECKS every
// word it received (a kernel after the exchange) and POISONS what it sent as soon as the transport's contract lets it
// (right after a non-deferred exchange; after mggcn_comm_release[_rank] with MGGCN_COMM_DEFER_RELEASE).
// Mutation runs drop one hipStreamWaitEvent at a time: most of them must make a check fail -- the model can tell.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "hipsim.h"
#include "mggcn_comm.h"

static std::atomic<long> g_bad{0}, g_checks{0};

// the one engine entry point comm.cpp calls (the all-reduce's local sum): a kernel on the stream
extern "C" void mggcn_axpy_f32(mggcn_stream_t stream, const float *A, float *B, float alpha, size_t size) {
    hipsim_enqueue(reinterpret_cast<hipStream_t>(stream), [A, B, alpha, size] {
        for (size_t i = 0; i < size; i++) B[i] += alpha * A[i];
    });
}

static float val(int rank, int it, int what, size_t i) { return (float)(((rank + 1) * 1000 + it) * 8 + what) + (float)i * 0.0009765625f; }
static const float kPoison = -12345.0f;

static void fill(hipStream_t s, float *p, size_t n, int rank, int it, int what) {
    hipsim_enqueue(s, [=] { for (size_t i = 0; i < n; i++) p[i] = val(rank, it, what, i); });
}
static void poison(hipStream_t s, float *p, size_t n) {
    hipsim_enqueue(s, [=] { for (size_t i = 0; i < n; i++) p[i] = kPoison; });
}
static void expect(hipStream_t s, const float *p, size_t n, int rank, int it, int what, const char *where, int me) {
    hipsim_enqueue(s, [=] {
        g_checks++;
        for (size_t i = 0; i < n; i++)
            if (p[i] != val(rank, it, what, i)) {
                if (g_bad++ < 5 && !std::getenv("COMM_SIM_QUIET"))
                    std::fprintf(stderr, "  %s: rank %d, piece of rank %d, iteration %d, word %zu: got %g, want %g\n", where, me, rank, it,
                                 i, (double)p[i], (double)val(rank, it, what, i));
                return;
            }
    });
}

struct scenario {
    int value_period;           // fills depend on iteration / value_period: > 1 when several iterations pass between two releases
                                // (a sender that has not released may only "overwrite" what it sent with the same values)
    int P;
    bool rccl;                  // the RCCL transport (on the model of rccl/rccl.h) instead of peer copies
    bool push, threaded;
    unsigned flags;
    int release_every;          // with MGGCN_COMM_DEFER_RELEASE: explicit release after every n-th exchange (poison only then)
    int iters;
    hipsim_policy policy;
    std::uint64_t seed;
    int peer_streams;           // MGGCN_P2P_PEER_STREAMS: -1 default (per peer between different devices), 0, 1
};

struct world {
    scenario sc;
    mggcn_comm *comm = nullptr;
    std::vector<hipStream_t> st;
    std::vector<mggcn_stream_t> st_abi;
    size_t count = 6;
    std::vector<std::vector<float>> send, bsend, recv, a2a_send, a2a_recv, red;     // (one send buffer per kind of exchange)
    std::vector<const float *> send_p, a2a_send_p;
    std::vector<float *> recv_p, a2a_recv_p, red_p;
    std::vector<int> exchanges;             // per rank (threaded) / [0] (one thread): exchanges since the last explicit release
};

static std::vector<size_t> a2a_counts(int P, int it) {
    std::vector<size_t> c((size_t)P * P);
    for (int j = 0; j < P; j++)
        for (int k = 0; k < P; k++) c[(size_t)j * P + k] = (size_t)((j * 7 + k * 3 + it) % 5);       // zeros included
    return c;
}

// may rank(s) overwrite what they sent now?  (non-deferred: always; deferred: after the explicit release this returns true for)
static bool release_now(world &w, int slot) {
    if (!(w.sc.flags & MGGCN_COMM_DEFER_RELEASE)) return true;
    if (++w.exchanges[slot] < w.sc.release_every) return false;
    w.exchanges[slot] = 0;
    return true;
}

// ---- one iteration, rank j, from rank j's own thread -------------------------------------------------------------------
static void iteration_rank(world &w, int j, int iter) {
    const int P = w.sc.P;
    const int it = iter / w.sc.value_period;
    const size_t n = w.count;
    hipStream_t s = w.st[j];
    const bool deferred = w.sc.flags & MGGCN_COMM_DEFER_RELEASE, skip_self = w.sc.flags & MGGCN_COMM_SKIP_SELF;
    // after every exchange: if the contract lets this rank overwrite what it has sent so far, it does
    auto after = [&] {
        if (!release_now(w, j)) return;
        if (deferred) mggcn_comm_release_rank(w.comm, j, w.st_abi[j]);
        poison(s, w.send[j].data(), n);
        poison(s, w.bsend[j].data(), n);
        poison(s, w.a2a_send[j].data(), w.a2a_send[j].size());
    };
    // all-gather
    fill(s, w.send[j].data(), n, j, it, 0);
    mggcn_comm_allgather_rank_f32(w.comm, j, w.send_p.data(), w.recv_p.data(), n, w.st_abi[j]);
    for (int i = 0; i < P; i++)
        if (i != j || !skip_self) expect(s, w.recv[j].data() + (size_t)i * n, n, i, it, 0, "all-gather", j);
    after();
    // broadcast from a moving root
    const int root = iter % P;
    if (j == root) fill(s, w.bsend[j].data(), n, j, it, 1);
    mggcn_comm_broadcast_rank_f32(w.comm, j, w.bsend[root].data(), w.recv_p.data(), n, root, w.st_abi[j]);
    expect(s, w.recv[j].data(), n, root, it, 1, "broadcast", j);
    after();
    // all-to-all with uneven pieces
    const auto counts = a2a_counts(P, it);
    std::vector<size_t> sdis((size_t)P * P), rdis((size_t)P * P);
    mggcn_comm_alltoallv_displacements(P, counts.data(), sdis.data(), rdis.data());
    for (int k = 0; k < P; k++) fill(s, w.a2a_send[j].data() + sdis[(size_t)j * P + k], counts[(size_t)j * P + k], j, it, 2 + k % 4);
    mggcn_comm_alltoallv_rank_f32(w.comm, j, w.a2a_send_p.data(), w.a2a_recv_p.data(), counts.data(), w.st_abi[j]);
    for (int k = 0; k < P; k++) expect(s, w.a2a_recv[j].data() + rdis[(size_t)j * P + k], counts[(size_t)k * P + j], k, it, 2 + j % 4, "all-to-all", j);
    after();
    // all-reduce (never deferred): small integers, the sum is exact and the same bits on every rank
    const size_t rn = 5;
    hipsim_enqueue(s, [p = w.red[j].data(), rn, j, it] { for (size_t i = 0; i < rn; i++) p[i] = (float)((j + 1) * (it + 1) + (int)i); });
    mggcn_comm_allreduce_sum_rank_f32(w.comm, j, w.red_p.data(), rn, w.st_abi[j]);
    hipsim_enqueue(s, [p = w.red[j].data(), rn, P, it, j] {
        g_checks++;
        for (size_t i = 0; i < rn; i++) {
            const float want = (float)((it + 1) * P * (P + 1) / 2 + (int)i * P);
            if (p[i] != want) { if (g_bad++ < 5 && !std::getenv("COMM_SIM_QUIET")) std::fprintf(stderr, "  all-reduce: rank %d word %zu: got %g, want %g\n", j, i, (double)p[i], (double)want); return; }
        }
    });
    after();
}

// ---- one iteration, all ranks, from ONE thread (the reference's process model: all-ranks entry points) -----------------
static void iteration_all(world &w, int iter) {
    const int P = w.sc.P;
    const int it = iter / w.sc.value_period;
    const size_t n = w.count;
    const bool deferred = w.sc.flags & MGGCN_COMM_DEFER_RELEASE, skip_self = w.sc.flags & MGGCN_COMM_SKIP_SELF;
    auto after = [&] {
        if (!release_now(w, 0)) return;
        if (deferred) mggcn_comm_release(w.comm, w.st_abi.data());
        for (int j = 0; j < P; j++) {
            poison(w.st[j], w.send[j].data(), n);
            poison(w.st[j], w.bsend[j].data(), n);
            poison(w.st[j], w.a2a_send[j].data(), w.a2a_send[j].size());
        }
    };
    for (int j = 0; j < P; j++) fill(w.st[j], w.send[j].data(), n, j, it, 0);
    mggcn_comm_allgather_f32(w.comm, w.send_p.data(), w.recv_p.data(), n, w.st_abi.data());
    for (int j = 0; j < P; j++)
        for (int i = 0; i < P; i++)
            if (i != j || !skip_self) expect(w.st[j], w.recv[j].data() + (size_t)i * n, n, i, it, 0, "all-gather", j);
    after();
    const int root = iter % P;
    fill(w.st[root], w.bsend[root].data(), n, root, it, 1);
    mggcn_comm_broadcast_f32(w.comm, w.bsend[root].data(), w.recv_p.data(), n, root, w.st_abi.data());
    for (int j = 0; j < P; j++) expect(w.st[j], w.recv[j].data(), n, root, it, 1, "broadcast", j);
    after();
    const auto counts = a2a_counts(P, it);
    std::vector<size_t> sdis((size_t)P * P), rdis((size_t)P * P);
    mggcn_comm_alltoallv_displacements(P, counts.data(), sdis.data(), rdis.data());
    for (int j = 0; j < P; j++)
        for (int k = 0; k < P; k++) fill(w.st[j], w.a2a_send[j].data() + sdis[(size_t)j * P + k], counts[(size_t)j * P + k], j, it, 2 + k % 4);
    mggcn_comm_alltoallv_f32(w.comm, w.a2a_send_p.data(), w.a2a_recv_p.data(), counts.data(), w.st_abi.data());
    for (int j = 0; j < P; j++)
        for (int k = 0; k < P; k++) expect(w.st[j], w.a2a_recv[j].data() + rdis[(size_t)j * P + k], counts[(size_t)k * P + j], k, it, 2 + j % 4, "all-to-all", j);
    after();
    const size_t rn = 5;
    for (int j = 0; j < P; j++)
        hipsim_enqueue(w.st[j], [p = w.red[j].data(), rn, j, it] { for (size_t i = 0; i < rn; i++) p[i] = (float)((j + 1) * (it + 1) + (int)i); });
    mggcn_comm_allreduce_sum_f32(w.comm, w.red_p.data(), rn, w.st_abi.data());
    for (int j = 0; j < P; j++)
        hipsim_enqueue(w.st[j], [p = w.red[j].data(), rn, P, it, j] {
            g_checks++;
            for (size_t i = 0; i < rn; i++) {
                const float want = (float)((it + 1) * P * (P + 1) / 2 + (int)i * P);
                if (p[i] != want) { if (g_bad++ < 5 && !std::getenv("COMM_SIM_QUIET")) std::fprintf(stderr, "  all-reduce: rank %d word %zu: got %g, want %g\n", j, i, (double)p[i], (double)want); return; }
            }
        });
    after();
}

// returns the number of failed checks
static long run(const scenario &sc, std::int64_t drop_wait = -1, std::uint64_t *waits = nullptr, int drop_class = HIPSIM_NO_CLASS) {
    hipsim_reset();
    hipsim_set_schedule(sc.seed, sc.policy);
    hipsim_drop_wait(drop_wait);
    hipsim_drop_class(drop_class);
    setenv("MGGCN_COMM_TRANSPORT", sc.rccl ? "rccl" : "p2p", 1);
    setenv("MGGCN_P2P_PUSH", sc.push ? "1" : "0", 1);
    if (sc.peer_streams < 0) unsetenv("MGGCN_P2P_PEER_STREAMS");
    else setenv("MGGCN_P2P_PEER_STREAMS", sc.peer_streams ? "1" : "0", 1);
    world w;
    w.sc = sc;
    const int P = sc.P;
    std::vector<int> devices(P);
    for (int j = 0; j < P; j++) devices[j] = j;                         // every rank on a GPU of its own
    w.comm = mggcn_comm_init_all(P, devices.data());
    mggcn_comm_set_exchange_flags(w.comm, sc.flags);
    const size_t a2a_max = (size_t)P * 5;
    for (int j = 0; j < P; j++) {
        w.st.push_back(hipsim_stream_create(j));
        w.st_abi.push_back(reinterpret_cast<mggcn_stream_t>(w.st.back()));
        w.send.emplace_back(w.count, kPoison);
        w.bsend.emplace_back(w.count, kPoison);
        w.recv.emplace_back((size_t)P * w.count, kPoison);
        w.a2a_send.emplace_back(a2a_max, kPoison);
        w.a2a_recv.emplace_back(a2a_max, kPoison);
        w.red.emplace_back(5, kPoison);
    }
    for (int j = 0; j < P; j++) {
        w.send_p.push_back(w.send[j].data()); w.recv_p.push_back(w.recv[j].data());
        w.a2a_send_p.push_back(w.a2a_send[j].data()); w.a2a_recv_p.push_back(w.a2a_recv[j].data());
        w.red_p.push_back(w.red[j].data());
    }
    w.exchanges.assign(P, 0);
    g_bad = 0; g_checks = 0;
    if (sc.threaded) {
        std::vector<std::thread> th;
        for (int j = 0; j < P; j++)
            th.emplace_back([&w, j] { for (int it = 0; it < w.sc.iters; it++) iteration_rank(w, j, it); });
        for (auto &t : th) t.join();
    } else {
        for (int it = 0; it < sc.iters; it++) iteration_all(w, it);
    }
    hipsim_drain();
    if (waits) *waits = hipsim_waits_seen();
    const long want_checks = (long)sc.iters * P * ((sc.flags & MGGCN_COMM_SKIP_SELF ? P - 1 : P) + 1 + P + 1);
    if (g_checks != want_checks) { std::fprintf(stderr, "  %ld checks ran, %ld expected\n", (long)g_checks, want_checks); g_bad++; }
    mggcn_comm_destroy(w.comm);
    return g_bad;
}

int main() {
    int failures = 0, scenarios = 0;
    const unsigned product = MGGCN_COMM_DEFER_RELEASE | MGGCN_COMM_SKIP_SELF;     // what host/dist_matrix.hpp sets
    for (const int P : {2, 3, 4, 8})
        for (const int transport : {0, 1, 2})                                       // receivers pull, senders push, RCCL
            for (const bool threaded : {true, false})
                for (const int variant : {0, 1, 2, 3}) {
                    if (transport == 2 && variant == 3) continue;                   // (no copying streams to vary)
                    scenario sc{};
                    const bool push = transport == 1;
                    sc.P = P; sc.push = push; sc.rccl = transport == 2; sc.threaded = threaded;
                    sc.flags = variant == 1 ? 0u : product;                         // 1: NCCL's contract, nothing deferred
                    sc.release_every = variant == 2 ? 32 : 1;                        // 2: releases are rare -> the forced release, ring re-use
                    sc.value_period = variant == 2 ? 8 : 1;                          //    (8 iterations x 4 exchanges between two releases)
                    sc.peer_streams = variant == 3 ? 0 : -1;                         // 3: one stream per rank instead of one per peer GPU
                    sc.iters = variant == 2 ? 40 : 12;                               // x 4 exchanges: the 32 event slots wrap
                    for (const std::uint64_t seed : {1u, 2u, 3u}) {
                        sc.seed = seed + 17u * (unsigned)P;
                        sc.policy = seed == 1 ? HIPSIM_RANDOM : seed == 2 ? HIPSIM_NEWEST_STREAM_FIRST : HIPSIM_OLDEST_STREAM_FIRST;
                        const long bad = run(sc);
                        scenarios++;
                        if (bad) {
                            failures++;
                            std::printf("TEST FAILED: P=%d %s %s flags=%u release_every=%d peer_streams=%d policy=%d: %ld bad\n", P,
                                        sc.rccl ? "rccl" : push ? "push" : "pull", threaded ? "threads" : "one thread", sc.flags, sc.release_every,
                                        sc.peer_streams, (int)sc.policy, bad);
                        }
                    }
                }
    std::printf("%s: %d scenarios of the two transports on the stream model, %d failed\n", failures ? "TEST FAILED" : "TEST PASSED",
                scenarios, failures);

    // can the model tell?  one thread (deterministic call order), one dropped hipStreamWaitEvent at a time
    setenv("COMM_SIM_QUIET", "1", 1);
    const int n_schedules = std::getenv("COMM_SIM_SCHEDULES") ? std::atoi(std::getenv("COMM_SIM_SCHEDULES")) : 12;
    for (const bool push : {false, true})
        for (const unsigned flags : {product, 0u}) {
            scenario sc{};
            sc.P = 4; sc.push = push; sc.threaded = false; sc.flags = flags; sc.release_every = 1; sc.value_period = 1; sc.peer_streams = -1; sc.iters = 3;
            sc.policy = HIPSIM_RANDOM; sc.seed = 5;
            std::uint64_t waits = 0;
            if (run(sc, -1, &waits)) { failures++; std::printf("TEST FAILED: mutation baseline\n"); continue; }
            int caught = 0, tried = 0;
            for (std::uint64_t k = 0; k < waits; k += 3) {
                bool seen = false;
                for (int schedule = 0; schedule < n_schedules && !seen; schedule++) {
                    sc.policy = schedule == 0 ? HIPSIM_NEWEST_STREAM_FIRST : schedule == 1 ? HIPSIM_OLDEST_STREAM_FIRST : HIPSIM_RANDOM;
                    sc.seed = 5 + (unsigned)schedule;
                    seen = run(sc, (std::int64_t)k) != 0;
                }
                tried++; caught += seen;
                if (std::getenv("COMM_SIM_MUTANTS")) std::printf("%s%llu", seen ? " +" : " -", (unsigned long long)k);
            }
            const bool ok = n_schedules < 40 || caught * 20 >= tried * 9;           // >= 45 % at 40 schedules (the others are implied by stream order, or have no successor to hurt)
            if (!ok) failures++;
            std::printf("%s: %s flags=%u: %d of %d single dropped waits (of %llu) make a check fail\n", ok ? "TEST PASSED" : "TEST FAILED",
                        push ? "push" : "pull", flags, caught, tried, (unsigned long long)waits);
        }
    // ... and one KIND of wait at a time, all of its instances: each kind must be missed in every form that has it
    for (const bool push : {false, true})
        for (const unsigned flags : {product, 0u})
            for (const int cls : {HIPSIM_CROSS_DEVICE, HIPSIM_JOIN, HIPSIM_FORK}) {
                scenario sc{};
                sc.P = 4; sc.push = push; sc.threaded = false; sc.flags = flags; sc.release_every = 1; sc.value_period = 1; sc.peer_streams = -1; sc.iters = 4;
                bool seen = false;
                for (int schedule = 0; schedule < std::max(n_schedules, 3) && !seen; schedule++) {
                    sc.policy = schedule == 0 ? HIPSIM_NEWEST_STREAM_FIRST : schedule == 1 ? HIPSIM_OLDEST_STREAM_FIRST : HIPSIM_RANDOM;
                    sc.seed = 11 + (unsigned)schedule;
                    seen = run(sc, -1, nullptr, cls) != 0;
                }
                if (!seen) failures++;
                std::printf("%s: %s flags=%u without its %s waits: %s\n", seen ? "TEST PASSED" : "TEST FAILED", push ? "push" : "pull", flags,
                            cls == HIPSIM_CROSS_DEVICE ? "cross-device" : cls == HIPSIM_JOIN ? "join" : "fork", seen ? "checks fail" : "NOT NOTICED");
            }
    std::printf("%s\n", failures ? "SOME FAILED" : "ALL PASSED");
    return failures ? 1 : 0;
}
