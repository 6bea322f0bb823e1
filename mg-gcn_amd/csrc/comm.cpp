// comm.cpp -- libmggcn_comm.so: the collectives of the single-process, P-GPU host layer.
// See include/mggcn_comm.h for the reference call sites this replaces.
//
// Two transports behind the same entry points:
//   rccl  one RCCL communicator per GPU (ncclCommInitAll), every collective a group of P
//         per-communicator calls -- the reference's NCCL pattern (src/dist_matrix.hpp:26-31,
//         :458-467, :587-592).  Default when the P ranks sit on P different GPUs.
//   p2p   plain device-to-device copies (hipMemcpyPeerAsync over xGMI, or same-device copies)
//         ordered by events: every receiver PULLS the pieces it needs on its own stream once
//         the senders' streams have produced them, and no sender runs on before every
//         receiver has read its buffer.  Chosen automatically when two ranks share a GPU
//         (RCCL refuses that) -- which is how the P > 1 schedules of the host layer are run
//         on a one-GPU box -- or with MGGCN_COMM_TRANSPORT=p2p.  Sums are formed in rank
//         order on every GPU: identical bits everywhere, like a ring all-reduce's result.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>
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

struct mggcn_comm {
    std::vector<ncclComm_t> comms;      // rccl transport; empty for p2p
    std::vector<int> devices;
    bool p2p = false;
    std::vector<hipEvent_t> ready, done;   // p2p: one pair per rank
    std::vector<float *> scratch;          // p2p all-reduce: P x count floats per rank
    std::vector<size_t> scratch_floats;
};

namespace {

inline hipStream_t as_stream(mggcn_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

void copy_f32(const mggcn_comm *c, float *dst, int dst_rank, const float *src, int src_rank, size_t count,
              hipStream_t st) {
    if (!count) return;
    const int dd = c->devices[dst_rank], sd = c->devices[src_rank];
    if (dd == sd) CHECK_HIP(hipMemcpyAsync(dst, src, count * sizeof(float), hipMemcpyDeviceToDevice, st));
    else CHECK_HIP(hipMemcpyPeerAsync(dst, dd, src, sd, count * sizeof(float), st));
}

// p2p skeleton: pull(j) enqueues rank j's copies on streams[j].  Before them stream j waits for
// every rank's earlier work (the data is produced on those streams); after them no stream
// proceeds until every rank has finished reading (the callers reuse their send buffers).
template <typename F>
void p2p_exchange(mggcn_comm *c, const mggcn_stream_t *streams, F &&pull) {
    const int P = (int)c->devices.size();
    for (int i = 0; i < P; i++) {
        CHECK_HIP(hipSetDevice(c->devices[i]));
        CHECK_HIP(hipEventRecord(c->ready[i], as_stream(streams[i])));
    }
    for (int j = 0; j < P; j++) {
        CHECK_HIP(hipSetDevice(c->devices[j]));
        for (int i = 0; i < P; i++)
            if (i != j) CHECK_HIP(hipStreamWaitEvent(as_stream(streams[j]), c->ready[i], 0));
        pull(j);
        CHECK_HIP(hipEventRecord(c->done[j], as_stream(streams[j])));
    }
    for (int i = 0; i < P; i++) {
        CHECK_HIP(hipSetDevice(c->devices[i]));
        for (int j = 0; j < P; j++)
            if (i != j) CHECK_HIP(hipStreamWaitEvent(as_stream(streams[i]), c->done[j], 0));
    }
}

}  // namespace

MGGCN_API mggcn_comm *mggcn_comm_init_all(int P, const int *devices) {
    if (P <= 0) {
        std::fprintf(stderr, "MGGCN precondition failed: communicator size must be positive\n");
        std::exit(EXIT_FAILURE);
    }
    auto *c = new mggcn_comm;
    c->devices.resize(P);
    for (int i = 0; i < P; i++) c->devices[i] = devices ? devices[i] : i;
    const std::set<int> distinct(c->devices.begin(), c->devices.end());
    const char *tr = std::getenv("MGGCN_COMM_TRANSPORT");
    c->p2p = (int)distinct.size() != P || (tr && std::strcmp(tr, "p2p") == 0);
    if (tr && std::strcmp(tr, "rccl") == 0 && (int)distinct.size() != P) {
        std::fprintf(stderr, "MGGCN precondition failed: MGGCN_COMM_TRANSPORT=rccl needs one GPU per rank\n");
        std::exit(EXIT_FAILURE);
    }
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
    c->ready.resize(P);
    c->done.resize(P);
    c->scratch.assign(P, nullptr);
    c->scratch_floats.assign(P, 0);
    for (int i = 0; i < P; i++) {
        CHECK_HIP(hipSetDevice(c->devices[i]));
        CHECK_HIP(hipEventCreateWithFlags(&c->ready[i], hipEventDisableTiming));
        CHECK_HIP(hipEventCreateWithFlags(&c->done[i], hipEventDisableTiming));
    }
    return c;
}

MGGCN_API void mggcn_comm_destroy(mggcn_comm *comm) {
    if (!comm) return;
    for (auto &c : comm->comms) ncclCommDestroy(c);
    for (size_t i = 0; i < comm->ready.size(); i++) {
        (void)hipSetDevice(comm->devices[i]);
        (void)hipEventDestroy(comm->ready[i]);
        (void)hipEventDestroy(comm->done[i]);
        if (comm->scratch[i]) (void)hipFree(comm->scratch[i]);
    }
    delete comm;
}

MGGCN_API int mggcn_comm_size(const mggcn_comm *comm) { return (int)comm->devices.size(); }

MGGCN_API const char *mggcn_comm_transport(const mggcn_comm *comm) { return comm->p2p ? "p2p" : "rccl"; }

MGGCN_API void mggcn_comm_broadcast_f32(mggcn_comm *comm, const float *send_root, float *const *recv,
                                        size_t count, int root, const mggcn_stream_t *streams) {
    const int P = (int)comm->devices.size();
    if (comm->p2p) {
        p2p_exchange(comm, streams, [&](int j) {
            if (recv[j] != send_root) copy_f32(comm, recv[j], j, send_root, root, count, as_stream(streams[j]));
        });
        return;
    }
    CHECK_RCCL(ncclGroupStart());
    for (int j = 0; j < P; j++) {
        CHECK_HIP(hipSetDevice(comm->devices[j]));
        CHECK_RCCL(ncclBroadcast(send_root, recv[j], count, ncclFloat32, root, comm->comms[j], as_stream(streams[j])));
    }
    CHECK_RCCL(ncclGroupEnd());
}

MGGCN_API void mggcn_comm_allgather_f32(mggcn_comm *comm, const float *const *send, float *const *recv,
                                        size_t count, const mggcn_stream_t *streams) {
    const int P = (int)comm->devices.size();
    if (comm->p2p) {
        p2p_exchange(comm, streams, [&](int j) {
            for (int i = 0; i < P; i++)
                if (recv[j] + (size_t)i * count != send[i])
                    copy_f32(comm, recv[j] + (size_t)i * count, j, send[i], i, count, as_stream(streams[j]));
        });
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

MGGCN_API void mggcn_comm_alltoallv_f32(mggcn_comm *comm, const float *const *send, float *const *recv,
                                        const size_t *counts, const mggcn_stream_t *streams) {
    const int P = (int)comm->devices.size();
    std::vector<size_t> sdis((size_t)P * P, 0), rdis((size_t)P * P, 0);
    mggcn_comm_alltoallv_displacements(P, counts, sdis.data(), rdis.data());
    if (comm->p2p) {
        p2p_exchange(comm, streams, [&](int k) {
            for (int j = 0; j < P; j++)
                copy_f32(comm, recv[k] + rdis[(size_t)k * P + j], k, send[j] + sdis[(size_t)j * P + k], j,
                         counts[(size_t)j * P + k], as_stream(streams[k]));
        });
        return;
    }
    CHECK_RCCL(ncclGroupStart());
    for (int j = 0; j < P; j++) {
        CHECK_HIP(hipSetDevice(comm->devices[j]));
        for (int k = 0; k < P; k++) {
            const size_t out = counts[(size_t)j * P + k], in = counts[(size_t)k * P + j];
            if (out) CHECK_RCCL(ncclSend(send[j] + sdis[(size_t)j * P + k], out, ncclFloat32, k, comm->comms[j], as_stream(streams[j])));
            if (in) CHECK_RCCL(ncclRecv(recv[j] + rdis[(size_t)j * P + k], in, ncclFloat32, k, comm->comms[j], as_stream(streams[j])));
        }
    }
    CHECK_RCCL(ncclGroupEnd());
}

MGGCN_API void mggcn_comm_allreduce_sum_f32(mggcn_comm *comm, float *const *bufs, size_t count,
                                            const mggcn_stream_t *streams) {
    const int P = (int)comm->devices.size();
    if (comm->p2p) {
        if (P == 1 || count == 0) return;
        for (int j = 0; j < P; j++)
            if (comm->scratch_floats[j] < (size_t)P * count) {
                CHECK_HIP(hipSetDevice(comm->devices[j]));
                CHECK_HIP(hipDeviceSynchronize());
                if (comm->scratch[j]) CHECK_HIP(hipFree(comm->scratch[j]));
                CHECK_HIP(hipMalloc(&comm->scratch[j], (size_t)P * count * sizeof(float)));
                comm->scratch_floats[j] = (size_t)P * count;
            }
        p2p_exchange(comm, streams, [&](int j) {
            for (int i = 0; i < P; i++)
                copy_f32(comm, comm->scratch[j] + (size_t)i * count, j, bufs[i], i, count, as_stream(streams[j]));
        });
        // every rank holds all P contributions and nobody reads bufs[] any more: sum in rank order
        for (int j = 0; j < P; j++) {
            CHECK_HIP(hipSetDevice(comm->devices[j]));
            CHECK_HIP(hipMemcpyAsync(bufs[j], comm->scratch[j], count * sizeof(float), hipMemcpyDeviceToDevice,
                                     as_stream(streams[j])));
            for (int i = 1; i < P; i++)
                mggcn_axpy_f32(streams[j], comm->scratch[j] + (size_t)i * count, bufs[j], 1.0f, count);
        }
        return;
    }
    CHECK_RCCL(ncclGroupStart());
    for (int j = 0; j < P; j++) {
        CHECK_HIP(hipSetDevice(comm->devices[j]));
        CHECK_RCCL(ncclAllReduce(bufs[j], bufs[j], count, ncclFloat32, ncclSum, comm->comms[j], as_stream(streams[j])));
    }
    CHECK_RCCL(ncclGroupEnd());
}
