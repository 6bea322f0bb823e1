// common.h -- shared helpers for the libmggcn_hip.so translation units (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "mggcn.h"

#define MGGCN_API extern "C" __attribute__((visibility("default")))

// Fail-fast like CHECK_CUDA (reference src/mg_gcn.hpp:31-39): message + exit.
#define MGGCN_CHECK_HIP(expr)                                                              \
    do {                                                                                   \
        hipError_t status_ = (expr);                                                       \
        if (status_ != hipSuccess) {                                                       \
            std::fprintf(stderr, "MGGCN HIP API failed at %s:%d with error: %s (%d)\n",    \
                         __FILE__, __LINE__, hipGetErrorString(status_), (int)status_);    \
            std::exit(EXIT_FAILURE);                                                       \
        }                                                                                  \
    } while (0)

// Preconditions the reference only asserts (src/cuda_utils.hpp:29).  A violated
// shape would become an out-of-bounds kernel, so these stay on in release builds.
#define MGGCN_REQUIRE(cond, msg)                                                           \
    do {                                                                                   \
        if (!(cond)) {                                                                     \
            std::fprintf(stderr, "MGGCN precondition failed at %s:%d: %s (%s)\n", __FILE__, \
                         __LINE__, msg, #cond);                                            \
            std::exit(EXIT_FAILURE);                                                       \
        }                                                                                  \
    } while (0)

#define MGGCN_CHECK_LAUNCH() MGGCN_CHECK_HIP(hipGetLastError())

static inline hipStream_t as_stream(mggcn_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

constexpr int kWave = 64;       // gfx950 wavefront
constexpr int kNumCU = 256;     // MI355X
constexpr int kNumXCD = 8;

static inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// grid for a grid-stride streaming kernel: enough workgroups to fill the chip
// (256 CUs x 8 blocks of 256 threads), never more than the work needs.
static inline unsigned stream_grid(size_t work_items, unsigned block = 256, unsigned per_thread = 1) {
    size_t blocks = (work_items + (size_t)block * per_thread - 1) / ((size_t)block * per_thread);
    if (blocks < 1) blocks = 1;
    const size_t cap = (size_t)kNumCU * 8;
    return (unsigned)(blocks < cap ? blocks : cap);
}
