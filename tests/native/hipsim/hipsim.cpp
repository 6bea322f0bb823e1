// tests/native/hipsim/hipsim.cpp -- the model described in hip/hip_runtime.h (test infrastructure, CPU only)
#include "hipsim.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <vector>

#include <rccl/rccl.h>

struct hipsim_event {
    std::uint64_t latest = 0;          // record instance this event stands for (0: never recorded)
};

namespace {
struct op {
    enum kind_t { kernel, wait, record } kind;
    std::uint64_t instance = 0;
    std::function<void()> fn;
};
}  // namespace

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
thread_local int t_device = 0;                             // hipSetDevice of the calling thread
hipsim_policy g_policy = HIPSIM_RANDOM;

std::uint64_t next_random() {                              // xorshift64*
    g_rng ^= g_rng >> 12; g_rng ^= g_rng << 25; g_rng ^= g_rng >> 27;
    return g_rng * 0x2545F4914F6CDD1Dull;
}

bool runnable(const hipsim_stream &s) {
    if (s.q.empty()) return false;
    const op &o = s.q.front();
    return o.kind != op::wait || g_reached[o.instance];
}

void drain_locked() {
    std::vector<hipsim_stream *> ready;
    for (;;) {
        ready.clear();
        bool pending = false;
        for (auto &s : g_streams) {
            if (!s->q.empty()) pending = true;
            if (runnable(*s)) ready.push_back(s.get());
        }
        if (!pending) return;
        if (ready.empty()) {
            std::fprintf(stderr, "hipsim: DEADLOCK -- every pending stream waits for an event record that cannot be reached\n");
            std::abort();
        }
        hipsim_stream *s = g_policy == HIPSIM_NEWEST_STREAM_FIRST ? ready.back()
                         : g_policy == HIPSIM_OLDEST_STREAM_FIRST ? ready.front()
                                                                  : ready[next_random() % ready.size()];
        // a random NUMBER of operations of that stream in a row: long runs of one stream are schedules too
        std::uint64_t burst = g_policy == HIPSIM_RANDOM ? 1 + next_random() % 4 : 1;
        while (burst-- && runnable(*s)) {
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
    s->q.push_back(op{op::record, e->latest, {}});
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
    if (!dropped && e->latest) s->q.push_back(op{op::wait, e->latest, {}});
    return hipSuccess;
}

void hipsim_enqueue(hipStream_t s, std::function<void()> kernel) {
    std::lock_guard<std::mutex> lk(g_mu);
    s->q.push_back(op{op::kernel, 0, std::move(kernel)});
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
void hipsim_drain() { std::lock_guard<std::mutex> lk(g_mu); drain_locked(); }
std::uint64_t hipsim_waits_seen() { std::lock_guard<std::mutex> lk(g_mu); return g_waits; }
void hipsim_drop_wait(std::int64_t k) { std::lock_guard<std::mutex> lk(g_mu); g_drop = k; }
void hipsim_drop_class(int cls) { std::lock_guard<std::mutex> lk(g_mu); g_drop_class = cls; }
std::uint64_t hipsim_ops_executed() { std::lock_guard<std::mutex> lk(g_mu); return g_executed; }
void hipsim_reset() {
    std::lock_guard<std::mutex> lk(g_mu);
    g_streams.clear(); g_events.clear(); g_reached.assign(1, 1); g_recorder.assign(1, nullptr);
    g_waits = g_executed = 0; g_drop = -1; g_drop_class = HIPSIM_NO_CLASS;
}

// ---- the RCCL transport has no model -----------------------------------------------------------
namespace { [[noreturn]] ncclResult_t no_rccl() { std::fprintf(stderr, "hipsim: the RCCL transport is not modelled\n"); std::abort(); } }
const char *ncclGetErrorString(ncclResult_t) { return "hipsim: no RCCL"; }
ncclResult_t ncclCommInitAll(ncclComm_t *, int, const int *) { no_rccl(); }
ncclResult_t ncclCommDestroy(ncclComm_t) { return ncclSuccess; }
ncclResult_t ncclGroupStart() { no_rccl(); }
ncclResult_t ncclGroupEnd() { no_rccl(); }
ncclResult_t ncclBroadcast(const void *, void *, std::size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) { no_rccl(); }
ncclResult_t ncclAllGather(const void *, void *, std::size_t, ncclDataType_t, ncclComm_t, hipStream_t) { no_rccl(); }
ncclResult_t ncclAllReduce(const void *, void *, std::size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) { no_rccl(); }
ncclResult_t ncclSend(const void *, std::size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) { no_rccl(); }
ncclResult_t ncclRecv(void *, std::size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) { no_rccl(); }
