// tests/native/hipsim/hipsim.h -- what a test drives the model of hip/hip_runtime.h with
#pragma once
#include <cstdint>
#include <functional>
#include <hip/hip_runtime.h>

hipStream_t hipsim_stream_create(int device);
void hipsim_enqueue(hipStream_t s, std::function<void()> kernel);      // a kernel: runs when the stream gets there
enum hipsim_policy { HIPSIM_RANDOM = 0, HIPSIM_NEWEST_STREAM_FIRST = 1, HIPSIM_OLDEST_STREAM_FIRST = 2 };
void hipsim_set_schedule(std::uint64_t seed, hipsim_policy policy);    // how drains pick among the runnable streams
void hipsim_drain();                                                   // execute everything enqueued so far
std::uint64_t hipsim_waits_seen();                                     // hipStreamWaitEvent calls so far
void hipsim_drop_wait(std::int64_t k);                                 // mutation: the k-th such call (0-based) is ignored; < 0: none
// mutation by KIND of wait: every wait of one class is ignored
//   CROSS_DEVICE  the event was recorded on a stream of another device (another rank's progress: ready / done / pushed)
//   JOIN          a test-made stream waits for a stream the code under test created on the same device (its copies have landed)
//   FORK          such a stream waits for the test-made stream of its device (the buffer it is about to touch is free / produced)
//   USER_EDGE     a test-made stream waits for ANOTHER test-made stream of the same device (the host layer's compute <-> comm edges)
enum { HIPSIM_NO_CLASS = 0, HIPSIM_CROSS_DEVICE = 1, HIPSIM_JOIN = 2, HIPSIM_FORK = 3, HIPSIM_USER_EDGE = 4 };
void hipsim_drop_class(int cls);
std::uint64_t hipsim_ops_executed();
void hipsim_reset();                                                   // forget every stream / event / counter (between scenarios)
