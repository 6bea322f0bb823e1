// mg_gcn.hpp -- ownership and error conventions of the C++17 host layer.
//
// Counterpart of the reference's src/mg_gcn.hpp (CHECK_* macros :31-68, cuda_ptr :71-72,
// cuda_malloc_managed :74-82, cuda_malloc :84-90) for a host that only ever talks to the
// device through the C ABI of include/mggcn.h: no HIP, CUDA, cuSPARSE, cuBLAS or NCCL header
// is visible from here on up.  Errors: the ABI itself prints file:line and exits, exactly like
// the reference's CHECK_CUDA; host-side problems throw and are caught in main() (main.cpp).
#pragma once

#include <cstddef>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "mggcn.h"

namespace mggcn {

// cuda_ptr<T>: shared ownership of a device allocation; sub-views alias the owner
// (reference src/mg_gcn.hpp:71-72 and the aliasing constructors of src/dist_matrix.hpp:155).
template <typename T>
using device_ptr = std::shared_ptr<T>;

template <typename T>
device_ptr<T> device_malloc(std::size_t count) {
    T *p = count ? static_cast<T *>(mggcn_malloc(count * sizeof(T))) : nullptr;
    return device_ptr<T>(p, [](T *q) { mggcn_free(q); });
}

// mapped pinned host memory (hipHostMalloc: visible to every device at the same address): for the few scalars a kernel
// reports to the host -- read after a synchronisation without a device-to-host copy
template <typename T>
device_ptr<T> host_malloc(std::size_t count) {
    T *p = count ? static_cast<T *>(mggcn_malloc_host(count * sizeof(T))) : nullptr;
    for (std::size_t i = 0; i < count; i++) p[i] = T();
    return device_ptr<T>(p, [](T *q) { mggcn_free_host(q); });
}

// view into an existing allocation, keeps the owner alive
template <typename T>
device_ptr<T> device_view(const device_ptr<T> &owner, std::size_t offset) {
    return device_ptr<T>(owner, owner.get() + offset);
}

// blocking host <-> device copies (the reference pokes managed memory directly; this pool
// has no XNACK, so host access is an explicit copy behind a device synchronise)
template <typename T>
void upload(T *dst_device, const T *src_host, std::size_t count) {
    mggcn_memcpy_h2d(dst_device, src_host, count * sizeof(T), nullptr);
    mggcn_stream_synchronize(nullptr);
}

template <typename T>
void download(T *dst_host, const T *src_device, std::size_t count) {
    mggcn_device_synchronize();
    mggcn_memcpy_d2h(dst_host, src_device, count * sizeof(T), nullptr);
    mggcn_stream_synchronize(nullptr);
}

}  // namespace mggcn

// the reference keeps these names at global scope
template <typename T>
using cuda_ptr = mggcn::device_ptr<T>;
