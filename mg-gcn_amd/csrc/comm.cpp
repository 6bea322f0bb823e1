// comm.cpp -- libmggcn_comm.so: single-process multi-GPU collectives over RCCL.
// See include/mggcn_comm.h for the reference call sites this replaces.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

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
    std::vector<ncclComm_t> comms;
    std::vector<int> devices;
};

MGGCN_API mggcn_comm *mggcn_comm_init_all(int P, const int *devices) {
    if (P <= 0) {
        std::fprintf(stderr, "MGGCN precondition failed: communicator size must be positive\n");
        std::exit(EXIT_FAILURE);
    }
    auto *c = new mggcn_comm;
    c->comms.resize(P);
    c->devices.resize(P);
    for (int i = 0; i < P; i++) c->devices[i] = devices ? devices[i] : i;
    CHECK_RCCL(ncclCommInitAll(c->comms.data(), P, c->devices.data()));
    return c;
}

MGGCN_API void mggcn_comm_destroy(mggcn_comm *comm) {
    if (!comm) return;
    for (auto &c : comm->comms) ncclCommDestroy(c);
    delete comm;
}

MGGCN_API int mggcn_comm_size(const mggcn_comm *comm) { return (int)comm->comms.size(); }

MGGCN_API void mggcn_comm_broadcast_f32(mggcn_comm *comm, const float *send_root, float *const *recv,
                                        size_t count, int root, const mggcn_stream_t *streams) {
    const int P = (int)comm->comms.size();
    CHECK_RCCL(ncclGroupStart());
    for (int j = 0; j < P; j++) {
        CHECK_HIP(hipSetDevice(comm->devices[j]));
        CHECK_RCCL(ncclBroadcast(send_root, recv[j], count, ncclFloat32, root, comm->comms[j],
                                 reinterpret_cast<hipStream_t>(streams[j])));
    }
    CHECK_RCCL(ncclGroupEnd());
}

MGGCN_API void mggcn_comm_allgather_f32(mggcn_comm *comm, const float *const *send, float *const *recv,
                                        size_t count, const mggcn_stream_t *streams) {
    const int P = (int)comm->comms.size();
    CHECK_RCCL(ncclGroupStart());
    for (int j = 0; j < P; j++) {
        CHECK_HIP(hipSetDevice(comm->devices[j]));
        CHECK_RCCL(ncclAllGather(send[j], recv[j], count, ncclFloat32, comm->comms[j],
                                 reinterpret_cast<hipStream_t>(streams[j])));
    }
    CHECK_RCCL(ncclGroupEnd());
}

MGGCN_API void mggcn_comm_allreduce_sum_f32(mggcn_comm *comm, float *const *bufs, size_t count,
                                            const mggcn_stream_t *streams) {
    const int P = (int)comm->comms.size();
    CHECK_RCCL(ncclGroupStart());
    for (int j = 0; j < P; j++) {
        CHECK_HIP(hipSetDevice(comm->devices[j]));
        CHECK_RCCL(ncclAllReduce(bufs[j], bufs[j], count, ncclFloat32, ncclSum, comm->comms[j],
                                 reinterpret_cast<hipStream_t>(streams[j])));
    }
    CHECK_RCCL(ncclGroupEnd());
}
