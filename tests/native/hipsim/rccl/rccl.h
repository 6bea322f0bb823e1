// tests/native/hipsim/rccl/rccl.h -- NOT RCCL: a MODEL of the part of its contract that mg-gcn_amd/csrc/comm.cpp relies on, on the
// streams of hip/hip_runtime.h (hipsim.cpp).  A call -- or everything between ncclGroupStart and ncclGroupEnd on one communicator --
// is ONE operation on the stream it was given; it runs when the matching operation of every rank it involves (all of them for a
// collective, the other side for a send / receive, matched by call order) has reached the head of ITS stream, and all of them
// run as one step: ranks that issue their collectives in different orders, or leave one out, hang -- reported as a deadlock.
// Counts and roots must agree; sums are formed in rank order.  fp32 only, one process, communicators from ncclCommInitAll.
#pragma once
#include <cstddef>
#include <hip/hip_runtime.h>
struct hipsim_nccl_comm;
typedef struct hipsim_nccl_comm *ncclComm_t;
typedef int ncclResult_t;
enum : int { ncclSuccess = 0 };
enum ncclDataType_t { ncclFloat32 = 7 };
enum ncclRedOp_t { ncclSum = 0 };
const char *ncclGetErrorString(ncclResult_t r);
ncclResult_t ncclCommInitAll(ncclComm_t *comms, int n, const int *devices);
ncclResult_t ncclCommDestroy(ncclComm_t c);
ncclResult_t ncclGroupStart();
ncclResult_t ncclGroupEnd();
ncclResult_t ncclBroadcast(const void *send, void *recv, std::size_t count, ncclDataType_t t, int root, ncclComm_t c, hipStream_t s);
ncclResult_t ncclAllGather(const void *send, void *recv, std::size_t count, ncclDataType_t t, ncclComm_t c, hipStream_t s);
ncclResult_t ncclAllReduce(const void *send, void *recv, std::size_t count, ncclDataType_t t, ncclRedOp_t op, ncclComm_t c, hipStream_t s);
ncclResult_t ncclSend(const void *send, std::size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t s);
ncclResult_t ncclRecv(void *recv, std::size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t s);
