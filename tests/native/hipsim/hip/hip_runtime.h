// tests/native/hipsim/hip/hip_runtime.h -- NOT HIP: a small MODEL of the stream / event semantics that
// mg-gcn_amd/csrc/comm.cpp (the peer-copy transport of libmggcn_comm.so) relies on, so that its host-side protocol can be
// compiled with plain g++ and run on the CPU, under the sanitizers and under adversarial schedules (tests/native/
// comm_sim_test.cpp).  Test infrastructure only; the product is never built against it.
//
// The model (hipsim.cpp):
//   * "device memory" is host memory; a stream is a FIFO of operations that nobody executes when they are enqueued;
//   * hipEventRecord(e, s) enqueues a marker on s and makes it e's LATEST record; hipStreamWaitEvent(s, e) enqueues "s goes on
//     once the record that was e's latest AT THE TIME OF THIS CALL has been reached" (a never-recorded event: no wait) --
//     the capture-at-call-time rule of the real API, which is what the transport's host sequence counters exist for;
//   * copies read their source and write their destination when they EXECUTE;
//   * hipsim_drain() executes everything enqueued so far, one operation at a time, always picking among the streams whose
//     head is allowed to run -- at random (seeded) or by a fixed adversarial preference.  An ordering the protocol forgot to
//     ask for is therefore free to go wrong; a cycle of waits is reported as a deadlock.
#pragma once
#include <cstddef>

typedef int hipError_t;
enum : int { hipSuccess = 0, hipErrorPeerAccessAlreadyEnabled = 704 };
typedef struct hipsim_stream *hipStream_t;
typedef struct hipsim_event *hipEvent_t;
enum hipMemcpyKind { hipMemcpyDeviceToDevice = 3 };
constexpr unsigned hipEventDisableTiming = 2u, hipStreamNonBlocking = 1u;

const char *hipGetErrorString(hipError_t e);
hipError_t hipGetLastError();
hipError_t hipSetDevice(int device);
hipError_t hipDeviceSynchronize();
hipError_t hipDeviceCanAccessPeer(int *can, int device, int peer);
hipError_t hipDeviceEnablePeerAccess(int peer, unsigned flags);
hipError_t hipDeviceGetStreamPriorityRange(int *least, int *greatest);
hipError_t hipsim_malloc(void **p, std::size_t bytes);
template <typename T> hipError_t hipMalloc(T **p, std::size_t bytes) { return hipsim_malloc(reinterpret_cast<void **>(p), bytes); }
hipError_t hipFree(void *p);
hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned flags, int priority);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned flags);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags);
hipError_t hipMemcpyAsync(void *dst, const void *src, std::size_t bytes, hipMemcpyKind kind, hipStream_t s);
hipError_t hipMemcpyPeerAsync(void *dst, int dst_device, const void *src, int src_device, std::size_t bytes, hipStream_t s);
