// runtime.hip -- device / stream / event / memory entry points of the C ABI.
// Stands in for the reference's per-GPU `context` (src/matrix.hpp:69-158) and the
// cuda_malloc helpers (src/mg_gcn.hpp:74-90).  Thin on purpose: the host layers
// (C++ headers, Python mirror) own all policy.
#include <algorithm>

#include "common.h"

MGGCN_API int mggcn_abi_version(void) { return MGGCN_ABI_VERSION; }

MGGCN_API int mggcn_device_count(void) {
    int n = 0;
    hipError_t st = hipGetDeviceCount(&n);
    if (st != hipSuccess) {  // no driver / no GPU: report zero devices, do not abort
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

MGGCN_API void mggcn_set_device(int device) { MGGCN_CHECK_HIP(hipSetDevice(device)); }

MGGCN_API int mggcn_get_device(void) {
    int d = 0;
    MGGCN_CHECK_HIP(hipGetDevice(&d));
    return d;
}

MGGCN_API void mggcn_device_synchronize(void) { MGGCN_CHECK_HIP(hipDeviceSynchronize()); }

MGGCN_API mggcn_stream_t mggcn_stream_create(int high_priority) {
    int least = 0, greatest = 0;  // numerically: greatest priority is the smaller number
    MGGCN_CHECK_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
    hipStream_t s = nullptr;
    MGGCN_CHECK_HIP(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, high_priority ? greatest : least));
    return s;
}

MGGCN_API void mggcn_stream_destroy(mggcn_stream_t stream) {
    if (!stream) return;
    mggcn_stream_release_scratch(stream);                  // per-stream reduction scratch (elementwise.hip)
    MGGCN_CHECK_HIP(hipStreamDestroy(as_stream(stream)));
}

MGGCN_API void mggcn_stream_synchronize(mggcn_stream_t stream) {
    MGGCN_CHECK_HIP(hipStreamSynchronize(as_stream(stream)));
}

MGGCN_API mggcn_event_t mggcn_event_create(void) {
    hipEvent_t e = nullptr;
    MGGCN_CHECK_HIP(hipEventCreate(&e));
    return e;
}

MGGCN_API void mggcn_event_destroy(mggcn_event_t event) {
    if (event) MGGCN_CHECK_HIP(hipEventDestroy(reinterpret_cast<hipEvent_t>(event)));
}

MGGCN_API void mggcn_event_record(mggcn_event_t event, mggcn_stream_t stream) {
    MGGCN_CHECK_HIP(hipEventRecord(reinterpret_cast<hipEvent_t>(event), as_stream(stream)));
}

MGGCN_API void mggcn_stream_wait_event(mggcn_stream_t stream, mggcn_event_t event) {
    MGGCN_CHECK_HIP(hipStreamWaitEvent(as_stream(stream), reinterpret_cast<hipEvent_t>(event), 0));
}

MGGCN_API void mggcn_event_synchronize(mggcn_event_t event) {
    MGGCN_CHECK_HIP(hipEventSynchronize(reinterpret_cast<hipEvent_t>(event)));
}

MGGCN_API float mggcn_event_elapsed_ms(mggcn_event_t begin, mggcn_event_t end) {
    float ms = 0.f;
    MGGCN_CHECK_HIP(hipEventElapsedTime(&ms, reinterpret_cast<hipEvent_t>(begin),
                                        reinterpret_cast<hipEvent_t>(end)));
    return ms;
}

MGGCN_API void *mggcn_malloc(size_t bytes) {
    void *p = nullptr;
    if (bytes) MGGCN_CHECK_HIP(hipMalloc(&p, bytes));
    return p;
}

MGGCN_API void mggcn_free(void *device_ptr) {
    if (device_ptr) MGGCN_CHECK_HIP(hipFree(device_ptr));
}

MGGCN_API void *mggcn_malloc_host(size_t bytes) {
    void *p = nullptr;
    if (bytes) MGGCN_CHECK_HIP(hipHostMalloc(&p, bytes, hipHostMallocDefault));
    return p;
}

MGGCN_API void mggcn_free_host(void *host_ptr) {
    if (host_ptr) MGGCN_CHECK_HIP(hipHostFree(host_ptr));
}

MGGCN_API void mggcn_memcpy_h2d(void *dst, const void *src, size_t bytes, mggcn_stream_t stream) {
    if (bytes) MGGCN_CHECK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, as_stream(stream)));
}

MGGCN_API void mggcn_memcpy_d2h(void *dst, const void *src, size_t bytes, mggcn_stream_t stream) {
    if (bytes) MGGCN_CHECK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, as_stream(stream)));
}

MGGCN_API void mggcn_memcpy_d2d(void *dst, const void *src, size_t bytes, mggcn_stream_t stream) {
    if (bytes) MGGCN_CHECK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, as_stream(stream)));
}

namespace {
__global__ __launch_bounds__(256) void zero_words_kernel(uint32_t *__restrict__ p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = 0u;
}

__global__ __launch_bounds__(256) void occupy_kernel(const volatile uint32_t *stop, unsigned long long max_ticks) {
    // one wave slot per SIMD of whatever CU the block lands on, like a communication kernel's channel; leaves when told to
    // or when the time is up (100 MHz constant clock): every wave reaches one of the two
    // (the flag is looked at every ~50 us: sixty-four waves polling a host-memory word back to back delay every packet the
    // command processor fetches over the same path -- the first version of this stand-in slowed the LAUNCHES it was meant to
    // share the device with by 20 us per workgroup of its own)
    // the clock is read after every sleep (~3.5 us: the kernel also serves as a calibrated delay, rank_epoch_model_r04.py)
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (unsigned it = 1; __builtin_amdgcn_s_memrealtime() - t0 < max_ticks; it++) {
        __builtin_amdgcn_s_sleep(127);
        if ((it & 15u) == 0 && *stop) break;
    }
}
}  // namespace

// Diagnostics (profiles/experiments/coresident_r04.py, never on the product path): n_blocks workgroups of 256 threads that
// do nothing but hold their wave slots until *stop_flag (device memory, or mapped pinned host memory) becomes non-zero or
// max_microseconds have passed -- a stand-in for the channels of a collective kernel that shares the device with the SpMM.
MGGCN_API void mggcn_debug_occupy_cus(mggcn_stream_t stream, uint32_t n_blocks, uint32_t max_microseconds, const uint32_t *stop_flag) {
    if (!n_blocks) return;
    hipLaunchKernelGGL(occupy_kernel, dim3(n_blocks), dim3(256), 0, as_stream(stream), stop_flag, (unsigned long long)max_microseconds * 100ull);
    MGGCN_CHECK_LAUNCH();
}

// Small word-aligned ranges (the loss layer's two scalars, a bias gradient) are zeroed by a kernel of this library, not by
// hipMemsetAsync: the runtime's fill path put a ~100 us bubble between its blit kernel and the next kernel of the stream
// (rocprofv3 kernel trace of an epoch, r02: fillBufferAligned -> softmax_xent 99 us on average, 10 us between ordinary
// kernels).  Works on device memory and on mapped pinned host memory (mggcn_malloc_host) alike.
MGGCN_API void mggcn_memset_zero(void *dst, size_t bytes, mggcn_stream_t stream) {
    if (!bytes) return;
    if (bytes <= (1u << 20) && bytes % 4 == 0 && (reinterpret_cast<uintptr_t>(dst) & 3u) == 0) {
        const size_t n = bytes / 4;
        hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 1024)), dim3(256), 0,
                           as_stream(stream), static_cast<uint32_t *>(dst), n);
        MGGCN_CHECK_LAUNCH();
        return;
    }
    MGGCN_CHECK_HIP(hipMemsetAsync(dst, 0, bytes, as_stream(stream)));
}
