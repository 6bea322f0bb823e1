// tests/native/hipsim/rccl/rccl.h -- NOT RCCL: declarations that let mg-gcn_amd/csrc/comm.cpp compile on the CPU for the model
// runs of its PEER-COPY transport (tests/native/comm_sim_test.cpp).  Every function aborts: the RCCL transport has no model here.
#pragma once
#include <cstddef>
#include <hip/hip_runtime.h>
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
