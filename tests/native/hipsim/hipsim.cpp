// tests/native/hipsim/hipsim.cpp -- the model described in hip/hip_runtime.h (test infrastructure, CPU only)
#include "hipsim.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <set>
#include <thread>
#include <vector>

#include <rccl/rccl.h>

struct hipsim_event {
    std::uint64_t latest = 0;          // record instance this event stands for (0: never recorded)
};

// ---- RCCL, modelled (see rccl/rccl.h): what one rank hands to the library in one call or one group ---------------------
struct hipsim_nccl_comm;
namespace {
struct nccl_prim {
    enum kind_t { allgather, broadcast, allreduce, send, recv } kind;
    const void *sendbuf;
    void *recvbuf;
    std::size_t count;          // floats
    int peer;                   // root of a broadcast, other side of a send / recv
    std::uint64_t seq;          // collectives: index of the call on this communicator; send / recv: index on this ordered pair
};
struct nccl_batch {
    hipsim_nccl_comm *comm;
    hipsim_stream *stream;
    std::vector<nccl_prim> prims;
};
struct op {
    enum kind_t { kernel, wait, record, batch } kind;
    std::uint64_t instance = 0;
    std::function<void()> fn;
    std::shared_ptr<nccl_batch> b;
};
}  // namespace

struct hipsim_nccl_clique;
struct hipsim_nccl_comm {
    hipsim_nccl_clique *clique = nullptr;
    int rank = 0;
    std::uint64_t coll_seq = 0;
    std::vector<std::uint64_t> send_seq, recv_seq;          // per peer
    std::deque<std::shared_ptr<nccl_batch>> queued;         // this rank's batches, in the order it issued them
    std::vector<nccl_prim> group;                           // calls of the open ncclGroup (issuing thread only)
    hipsim_stream *group_stream = nullptr;
};
struct hipsim_nccl_clique {
    std::vector<std::unique_ptr<hipsim_nccl_comm>> comms;
};

struct hipsim_stream {
    int device = 0;
    bool user = false;                 // created by the test (a rank's own stream), not by the code under test
    std::deque<op> q;
};

namespace {
std::mutex g_mu;                                           // enqueue threads and drains
std::vector<std::unique_ptr<hipsim_stream>> g_streams;
std::vector<std::unique_ptr<hipsim_event>> g_events;
std::vector<char> g_reached{1};                            // per record instance; instance 0 ("never recorded") counts as reached
std::uint64_t g_waits = 0, g_executed = 0, g_rng = 0x9E3779B97F4A7C15ull;
std::int64_t g_drop = -1;
int g_drop_class = HIPSIM_NO_CLASS;
std::vector<hipsim_stream *> g_recorder{nullptr};          // per record instance: the stream it was recorded on
std::vector<std::unique_ptr<hipsim_nccl_clique>> g_cliques;
thread_local int t_device = 0;                             // hipSetDevice of the calling thread
thread_local int t_group_depth = 0;                        // ncclGroupStart nesting of the calling thread
thread_local std::vector<hipsim_nccl_comm *> t_group_comms;
hipsim_policy g_policy = HIPSIM_RANDOM;

std::uint64_t next_random() {                              // xorshift64*
    g_rng ^= g_rng >> 12; g_rng ^= g_rng << 25; g_rng ^= g_rng >> 27;
    return g_rng * 0x2545F4914F6CDD1Dull;
}

// A batch of rank r can run when every primitive in it finds its counterpart in a batch that sits at the HEAD of its own
// stream -- the all-gather / broadcast / all-reduce of the same index on EVERY other rank, the receive (send) of the same
// index on the other side of a send (receive) -- and those batches can run too: the set is collected, executed as one step and
// popped together.  (What real collectives do in time -- all ranks inside the operation at once -- and what makes a wrong order
// of calls across ranks hang.)
const nccl_prim *find_match(const nccl_batch &pb, const nccl_prim &p, int my_rank) {
    for (const auto &q : pb.prims) {
        if (p.kind == nccl_prim::send) { if (q.kind == nccl_prim::recv && q.peer == my_rank && q.seq == p.seq) return &q; }
        else if (p.kind == nccl_prim::recv) { if (q.kind == nccl_prim::send && q.peer == my_rank && q.seq == p.seq) return &q; }
        else if (q.kind == p.kind && q.seq == p.seq) return &q;
    }
    return nullptr;
}

bool collect(nccl_batch *b, std::vector<nccl_batch *> &set) {
    if (std::find(set.begin(), set.end(), b) != set.end()) return true;
    if (b->stream->q.empty() || b->stream->q.front().kind != op::batch || b->stream->q.front().b.get() != b) return false;
    set.push_back(b);
    for (const auto &p : b->prims) {
        auto &comms = b->comm->clique->comms;
        for (int r = 0; r < (int)comms.size(); r++) {
            if (r == b->comm->rank) continue;
            if ((p.kind == nccl_prim::send || p.kind == nccl_prim::recv) && r != p.peer) continue;
            hipsim_nccl_comm &pc = *comms[r];
            if (pc.queued.empty()) return false;                                   // that rank has not issued it yet
            nccl_batch *pb = pc.queued.front().get();
            if (!find_match(*pb, p, b->comm->rank)) return false;                  // its next batch is something else
            if (!collect(pb, set)) return false;
        }
    }
    return true;
}

void execute(const std::vector<nccl_batch *> &set) {
    for (nccl_batch *b : set)
        for (const auto &p : b->prims) {
            auto &comms = b->comm->clique->comms;
            const int P = (int)comms.size(), me = b->comm->rank;
            auto other = [&](int r) { return find_match(*comms[r]->queued.front(), p, me); };
            switch (p.kind) {
                case nccl_prim::send: {                                            // the pair is executed from the sending side
                    const nccl_prim *q = other(p.peer);
                    if (q->count != p.count) { std::fprintf(stderr, "hipsim: send of %zu floats meets a receive of %zu\n", p.count, q->count); std::abort(); }
                    std::memmove(q->recvbuf, p.sendbuf, p.count * sizeof(float));
                    break;
                }
                case nccl_prim::recv: break;
                case nccl_prim::allgather:                                          // every rank fills its own receive buffer
                    for (int r = 0; r < P; r++) {
                        const nccl_prim *q = r == me ? &p : other(r);
                        if (q->count != p.count) { std::fprintf(stderr, "hipsim: all-gather counts differ\n"); std::abort(); }
                        std::memmove((float *)p.recvbuf + (std::size_t)r * p.count, q->sendbuf, p.count * sizeof(float));
                    }
                    break;
                case nccl_prim::broadcast: {
                    const nccl_prim *q = p.peer == me ? &p : other(p.peer);
                    if (q->peer != p.peer || q->count != p.count) { std::fprintf(stderr, "hipsim: broadcast root / count differ\n"); std::abort(); }
                    std::memmove(p.recvbuf, q->sendbuf, p.count * sizeof(float));
                    break;
                }
                case nccl_prim::allreduce: break;                                   // below: needs every input before any output
            }
        }
    // all-reduce (possibly in place): sums formed once, in rank order, from the inputs as they are now
    for (nccl_batch *b : set)
        for (const auto &p : b->prims) {
            if (p.kind != nccl_prim::allreduce || b->comm->rank != 0) continue;
            auto &comms = b->comm->clique->comms;
            std::vector<float> sum(p.count, 0.f);
            std::vector<const nccl_prim *> parts;
            for (int r = 0; r < (int)comms.size(); r++) parts.push_back(r == 0 ? &p : find_match(*comms[r]->queued.front(), p, 0));
            for (const nccl_prim *q : parts) {
                if (q->count != p.count) { std::fprintf(stderr, "hipsim: all-reduce counts differ\n"); std::abort(); }
                for (std::size_t i = 0; i < p.count; i++) sum[i] += ((const float *)q->sendbuf)[i];
            }
            for (const nccl_prim *q : parts) std::memcpy(q->recvbuf, sum.data(), p.count * sizeof(float));
        }
    for (nccl_batch *b : set) {                                                     // (b stays alive through the queue's shared_ptr until here)
        hipsim_nccl_comm *c = b->comm;
        b->stream->q.pop_front();
        c->queued.pop_front();
        g_executed++;
    }
}

bool runnable(hipsim_stream &s, std::vector<nccl_batch *> *set = nullptr) {
    if (s.q.empty()) return false;
    const op &o = s.q.front();
    if (o.kind == op::wait) return g_reached[o.instance] != 0;
    if (o.kind != op::batch) return true;
    std::vector<nccl_batch *> local;
    std::vector<nccl_batch *> &use = set ? *set : local;
    use.clear();
    return collect(o.b.get(), use);
}

void drain_locked(std::unique_lock<std::mutex> &lk) {
    std::vector<hipsim_stream *> ready;
    std::vector<nccl_batch *> set;
    auto stuck_since = std::chrono::steady_clock::time_point{};
    for (;;) {
        ready.clear();
        bool pending = false, batch_head = false;
        for (auto &s : g_streams) {
            if (!s->q.empty()) { pending = true; batch_head = batch_head || s->q.front().kind == op::batch; }
            if (runnable(*s)) ready.push_back(s.get());
        }
        if (!pending) return;
        if (ready.empty()) {
            // a collective whose other ranks have not been ISSUED yet (their enqueue threads are still on their way) is not a
            // deadlock: wait for the host, a while
            const auto now = std::chrono::steady_clock::now();
            if (batch_head && stuck_since == std::chrono::steady_clock::time_point{}) stuck_since = now;
            if (batch_head && now - stuck_since < std::chrono::seconds(20)) {
                lk.unlock();
                std::this_thread::sleep_for(std::chrono::milliseconds(1));
                lk.lock();
                continue;
            }
            std::fprintf(stderr, "hipsim: DEADLOCK -- every pending stream waits for an event record that cannot be reached%s\n",
                         batch_head ? " or for a collective the other ranks never issue in this order" : "");
            std::abort();
        }
        stuck_since = std::chrono::steady_clock::time_point{};
        hipsim_stream *s = g_policy == HIPSIM_NEWEST_STREAM_FIRST ? ready.back()
                         : g_policy == HIPSIM_OLDEST_STREAM_FIRST ? ready.front()
                                                                  : ready[next_random() % ready.size()];
        // a random NUMBER of operations of that stream in a row: long runs of one stream are schedules too
        std::uint64_t burst = g_policy == HIPSIM_RANDOM ? 1 + next_random() % 4 : 1;
        while (burst-- && runnable(*s, &set)) {
            if (s->q.front().kind == op::batch) { execute(set); continue; }
            op o = std::move(s->q.front());
            s->q.pop_front();
            if (o.kind == op::kernel) o.fn();
            else if (o.kind == op::record) g_reached[o.instance] = 1;
            g_executed++;
        }
    }
}
}  // namespace

const char *hipGetErrorString(hipError_t) { return "hipsim error"; }
hipError_t hipGetLastError() { return hipSuccess; }
hipError_t hipSetDevice(int device) { t_device = device; return hipSuccess; }
hipError_t hipDeviceSynchronize() { hipsim_drain(); return hipSuccess; }
hipError_t hipDeviceCanAccessPeer(int *can, int, int) { *can = 1; return hipSuccess; }
hipError_t hipDeviceEnablePeerAccess(int, unsigned) { return hipSuccess; }
hipError_t hipDeviceGetStreamPriorityRange(int *least, int *greatest) { *least = 0; *greatest = -1; return hipSuccess; }
hipError_t hipsim_malloc(void **p, std::size_t bytes) { *p = std::malloc(bytes ? bytes : 1); return *p ? hipSuccess : 2; }
hipError_t hipFree(void *p) { std::free(p); return hipSuccess; }

hipStream_t hipsim_stream_create(int device) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_streams.push_back(std::make_unique<hipsim_stream>());
    g_streams.back()->device = device;
    g_streams.back()->user = true;
    return g_streams.back().get();
}
hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned, int) {      // a stream of the code under test, on the current device
    std::lock_guard<std::mutex> lk(g_mu);
    g_streams.push_back(std::make_unique<hipsim_stream>());
    g_streams.back()->device = t_device;
    *s = g_streams.back().get();
    return hipSuccess;
}
hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }      // (kept until hipsim_reset: a destroyed stream still drains)
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_events.push_back(std::make_unique<hipsim_event>());
    *e = g_events.back().get();
    return hipSuccess;
}
hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }

hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_reached.push_back(0);
    g_recorder.push_back(s);
    e->latest = g_reached.size() - 1;
    s->q.push_back(op{op::record, e->latest, {}, {}});
    return hipSuccess;
}

hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned) {
    std::lock_guard<std::mutex> lk(g_mu);
    bool dropped = (std::int64_t)g_waits == g_drop;
    g_waits++;
    if (e->latest && g_drop_class != HIPSIM_NO_CLASS) {
        const hipsim_stream *from = g_recorder[e->latest];
        const int cls = from->device != s->device ? HIPSIM_CROSS_DEVICE
                      : (s->user && !from->user)  ? HIPSIM_JOIN
                      : (!s->user && from->user)  ? HIPSIM_FORK
                      : (s != from && s->user)    ? HIPSIM_USER_EDGE
                                                  : HIPSIM_NO_CLASS;
        dropped = dropped || cls == g_drop_class;
    }
    if (!dropped && e->latest) s->q.push_back(op{op::wait, e->latest, {}, {}});
    return hipSuccess;
}

void hipsim_enqueue(hipStream_t s, std::function<void()> kernel) {
    std::lock_guard<std::mutex> lk(g_mu);
    s->q.push_back(op{op::kernel, 0, std::move(kernel), {}});
}

hipError_t hipMemcpyAsync(void *dst, const void *src, std::size_t bytes, hipMemcpyKind, hipStream_t s) {
    hipsim_enqueue(s, [dst, src, bytes] { std::memmove(dst, src, bytes); });
    return hipSuccess;
}
hipError_t hipMemcpyPeerAsync(void *dst, int, const void *src, int, std::size_t bytes, hipStream_t s) {
    return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s);
}

void hipsim_set_schedule(std::uint64_t seed, hipsim_policy policy) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_rng = seed * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull;
    g_policy = policy;
}
void hipsim_drain() { std::unique_lock<std::mutex> lk(g_mu); drain_locked(lk); }
std::uint64_t hipsim_waits_seen() { std::lock_guard<std::mutex> lk(g_mu); return g_waits; }
void hipsim_drop_wait(std::int64_t k) { std::lock_guard<std::mutex> lk(g_mu); g_drop = k; }
void hipsim_drop_class(int cls) { std::lock_guard<std::mutex> lk(g_mu); g_drop_class = cls; }
std::uint64_t hipsim_ops_executed() { std::lock_guard<std::mutex> lk(g_mu); return g_executed; }
void hipsim_reset() {
    std::lock_guard<std::mutex> lk(g_mu);
    g_streams.clear(); g_events.clear(); g_cliques.clear(); g_reached.assign(1, 1); g_recorder.assign(1, nullptr);
    g_waits = g_executed = 0; g_drop = -1; g_drop_class = HIPSIM_NO_CLASS;
}

// ---- RCCL, modelled: enqueue-only like the real one; the data moves when the batches meet (collect / execute above) -------
namespace {
void post(ncclComm_t c, nccl_prim p, hipStream_t s) {
    if (t_group_depth > 0) {
        if (c->group.empty()) { c->group_stream = s; t_group_comms.push_back(c); }
        if (c->group_stream != s) { std::fprintf(stderr, "hipsim: one group, one communicator, two streams\n"); std::abort(); }
        c->group.push_back(p);
        return;
    }
    std::lock_guard<std::mutex> lk(g_mu);
    auto b = std::make_shared<nccl_batch>(nccl_batch{c, s, {p}});
    c->queued.push_back(b);
    s->q.push_back(op{op::batch, 0, {}, b});
}
}  // namespace
const char *ncclGetErrorString(ncclResult_t) { return "hipsim rccl error"; }
ncclResult_t ncclCommInitAll(ncclComm_t *comms, int n, const int *) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_cliques.push_back(std::make_unique<hipsim_nccl_clique>());
    auto &cl = *g_cliques.back();
    for (int r = 0; r < n; r++) {
        cl.comms.push_back(std::make_unique<hipsim_nccl_comm>());
        cl.comms.back()->clique = &cl; cl.comms.back()->rank = r;
        cl.comms.back()->send_seq.assign(n, 0); cl.comms.back()->recv_seq.assign(n, 0);
        comms[r] = cl.comms.back().get();
    }
    return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t) { return ncclSuccess; }      // (kept until hipsim_reset)
ncclResult_t ncclGroupStart() { t_group_depth++; return ncclSuccess; }
ncclResult_t ncclGroupEnd() {
    if (--t_group_depth > 0) return ncclSuccess;
    std::lock_guard<std::mutex> lk(g_mu);
    for (ncclComm_t c : t_group_comms) {
        auto b = std::make_shared<nccl_batch>(nccl_batch{c, c->group_stream, std::move(c->group)});
        c->group.clear();
        c->queued.push_back(b);
        c->group_stream->q.push_back(op{op::batch, 0, {}, b});
    }
    t_group_comms.clear();
    return ncclSuccess;
}
ncclResult_t ncclBroadcast(const void *send, void *recv, std::size_t count, ncclDataType_t, int root, ncclComm_t c, hipStream_t s) {
    post(c, nccl_prim{nccl_prim::broadcast, send, recv, count, root, c->coll_seq++}, s); return ncclSuccess;
}
ncclResult_t ncclAllGather(const void *send, void *recv, std::size_t count, ncclDataType_t, ncclComm_t c, hipStream_t s) {
    post(c, nccl_prim{nccl_prim::allgather, send, recv, count, -1, c->coll_seq++}, s); return ncclSuccess;
}
ncclResult_t ncclAllReduce(const void *send, void *recv, std::size_t count, ncclDataType_t, ncclRedOp_t, ncclComm_t c, hipStream_t s) {
    post(c, nccl_prim{nccl_prim::allreduce, send, recv, count, -1, c->coll_seq++}, s); return ncclSuccess;
}
ncclResult_t ncclSend(const void *send, std::size_t count, ncclDataType_t, int peer, ncclComm_t c, hipStream_t s) {
    post(c, nccl_prim{nccl_prim::send, send, nullptr, count, peer, c->send_seq[peer]++}, s); return ncclSuccess;
}
ncclResult_t ncclRecv(void *recv, std::size_t count, ncclDataType_t, int peer, ncclComm_t c, hipStream_t s) {
    post(c, nccl_prim{nccl_prim::recv, nullptr, recv, count, peer, c->recv_seq[peer]++}, s); return ncclSuccess;
}
