"""Host mirror of the reference's runtime and matrix types (src/matrix.hpp).

``context``     per-GPU streams / named events / timers       (src/matrix.hpp:69-158)
``csr_matrix``  CSR<u32,u32,f32> with a host copy + device copy (src/matrix.hpp:214-468)
``dn_matrix``   row-major dense matrix on the device           (src/matrix.hpp:478-639)

Same member names and meaning as the reference classes.  Differences forced by
the platform: the reference keeps everything in cudaMallocManaged memory and
pokes it from the host (``operator[]``); this pool has no XNACK, so matrices
live in device memory (torch tensors -- plumbing only) and host access is an
explicit ``numpy()`` / ``from_numpy()`` copy.  Every computation goes through
the C ABI in ``_lib`` -- torch never computes anything here.
"""
from __future__ import annotations

import os
from typing import Dict, Optional, Tuple

import numpy as np

from . import _lib
from . import datasets


def _torch():
    import torch
    return torch


class matrix_error(RuntimeError):
    """reference src/matrix.hpp:32-37"""


class context:
    """reference src/matrix.hpp:69-158.  cuda_streams[0] is the low-priority compute
    stream every op is enqueued on, cuda_streams[1] the high-priority communication
    stream (src/matrix.hpp:53-60, :82).  Events are named and created lazily."""

    def __init__(self, index: int = 0):
        _lib.require_gpu()
        torch = _torch()
        self.rank = int(index)
        self.lib = _lib.load()
        self.device = torch.device("cuda", self.rank)
        self.set()
        lo, hi = 0, -1   # torch: lower number = higher priority
        self.cuda_streams = [torch.cuda.Stream(device=self.device, priority=lo),
                             torch.cuda.Stream(device=self.device, priority=hi)]
        self.events: Dict[str, int] = {}
        self.timers: Dict[str, Tuple[str, str]] = {}
        self._workspace = None

    def __del__(self):
        # the library keeps a small reduction scratch per (device, stream): hand it back with the streams, so that a
        # recycled stream handle never inherits it (torch owns the streams; the ABI's own streams do this on destroy)
        # (looked up by stream alone: the thread's current device is left as it is -- this runs at garbage-collection
        # time, possibly in the middle of another context's work on another device)
        try:
            for st in self.cuda_streams:
                self.lib.mggcn_stream_release_scratch(st.cuda_stream)
        except Exception:
            pass

    # -- device / stream ----------------------------------------------------
    def set(self) -> None:
        self.lib.mggcn_set_device(self.rank)
        _torch().cuda.set_device(self.rank)

    def sync(self) -> None:
        self.set()
        self.lib.mggcn_device_synchronize()

    def stream(self, stream_id: int = 0) -> int:
        """raw hipStream_t handle for the C ABI"""
        return self.cuda_streams[stream_id].cuda_stream

    # -- named events ---------------------------------------------------------
    def record(self, name: str, stream_id: int) -> None:
        ev = self.events.get(name)
        if ev is None:
            ev = self.events[name] = self.lib.mggcn_event_create()
        self.lib.mggcn_event_record(ev, self.stream(stream_id))

    def wait(self, name: str, stream_id: int) -> None:
        self.lib.mggcn_stream_wait_event(self.stream(stream_id), self.events[name])

    def register_timer(self, name: str, beg: str, end: str) -> None:
        self.timers[name] = (beg, end)

    def measure(self, name: str) -> float:
        if name not in self.timers:
            return 0.0
        beg, end = self.timers[name]
        return float(self.lib.mggcn_event_elapsed_ms(self.events[beg], self.events[end]))

    def dump_timers(self, out, prefix: str = "") -> None:
        """``<prefix><name>:<ms>`` per line, sorted by name (std::map order),
        reference src/matrix.hpp:150-157."""
        for name in sorted(self.timers):
            out.write(f"{prefix}{name}:{self.measure(name):g}\n")

    def fill(self, mat: "dn_matrix", value: float) -> None:
        """constant fill ordered on the compute stream (torch is only the memset here)"""
        torch = _torch()
        with torch.cuda.stream(self.cuda_streams[0]):
            mat.t.fill_(value)

    # -- scratch for split-K GEMMs (cuBLAS keeps its own inside the handle) ----
    def workspace(self, nbytes: int):
        if nbytes == 0:
            return None
        if self._workspace is None or self._workspace.numel() < nbytes:
            self._workspace = _torch().empty(nbytes, dtype=_torch().uint8, device=self.device)
        return self._workspace


class host_scalars:
    """A few float32 scalars in MAPPED PINNED host memory (mggcn_malloc_host): kernels write them through the same
    address, the host reads them after a synchronisation without a device-to-host copy (the epoch's loss / accuracy:
    a `tensor.cpu()` there cost a blit kernel and ~70 us of idle GPU per epoch).  Quacks like the slice of a torch
    tensor where the ops layer needs it: data_ptr(), [a:b]."""

    def __init__(self, n: int, _base=None, _offset: int = 0):
        import ctypes
        self._n, self._offset = n, _offset
        if _base is None:
            lib = _lib.load()
            ptr = lib.mggcn_malloc_host(4 * n)
            if not ptr:
                raise MemoryError("mggcn_malloc_host")
            self._owner = self
            self._ptr, self._lib = ptr, lib
            self._view = np.ctypeslib.as_array((ctypes.c_float * n).from_address(ptr))
            self._view[:] = 0.0
        else:
            self._owner = _base
            self._ptr, self._lib = _base._ptr, None
            self._view = _base._view

    def data_ptr(self) -> int:
        return self._ptr + 4 * self._offset

    def __getitem__(self, key):
        if isinstance(key, slice):
            a, b, _ = key.indices(self._n)
            return host_scalars(b - a, _base=self._owner, _offset=self._offset + a)
        raise TypeError("host_scalars supports slices only")

    def numpy(self) -> np.ndarray:
        """a COPY of the current values (call after synchronising the stream that writes them)"""
        return np.array(self._view[self._offset:self._offset + self._n], dtype=np.float32)

    def __del__(self):
        if getattr(self, "_lib", None) is not None and self._ptr:
            try:
                self._lib.mggcn_free_host(self._ptr)
            except Exception:
                pass
            self._ptr = 0


class dn_matrix:
    """Row-major dense matrix, reference src/matrix.hpp:478-639.  ``dtype`` is
    float32 (r_t) or int32 (labels).  ``buffer`` shares storage like the
    reference's aliasing cuda_ptr constructor."""

    def __init__(self, n, m=None, buffer=None, dtype=np.float32, device=None):
        torch = _torch()
        if isinstance(n, (str, os.PathLike)):
            arr = self._read(os.fspath(n), dtype)
            n, m = arr.shape
            self.N_, self.M_ = int(n), int(m)
            self.t = torch.from_numpy(np.ascontiguousarray(arr)).to(device or "cuda")
            torch.cuda.current_stream(self.t.device).synchronize()
            return
        if isinstance(n, tuple):
            n, m = n
        self.N_, self.M_ = int(n), int(m)
        tdt = torch.float32 if np.dtype(dtype) == np.float32 else torch.int32
        if buffer is not None:
            assert buffer.numel() >= self.N_ * self.M_, "aliased buffer too small"
            self.t = buffer.view(-1)[: self.N_ * self.M_].view(self.N_, self.M_)
            assert self.t.dtype == tdt
        else:
            self.t = torch.empty((self.N_, self.M_), dtype=tdt, device=device or "cuda")

    @staticmethod
    def _read(path: str, dtype) -> np.ndarray:
        if not path.endswith(".bin"):
            raise matrix_error("File type is not supported.")
        try:
            return datasets.read_dense(path, "<f4" if np.dtype(dtype) == np.float32 else "<i4")
        except datasets.format_error as e:
            raise matrix_error(str(e))

    @classmethod
    def from_numpy(cls, arr: np.ndarray, device=None) -> "dn_matrix":
        arr = np.ascontiguousarray(arr)
        if arr.ndim == 1:
            arr = arr.reshape(-1, 1)
        dt = np.float32 if arr.dtype.kind == "f" else np.int32
        out = cls(arr.shape[0], arr.shape[1], dtype=dt, device=device)
        out.t.copy_(_torch().from_numpy(arr.astype(dt, copy=False)))
        _torch().cuda.current_stream(out.t.device).synchronize()   # visible to every stream afterwards
        return out

    def numpy(self) -> np.ndarray:
        return self.t.detach().cpu().numpy()

    def n(self) -> int: return self.N_
    def m(self) -> int: return self.M_
    def size(self) -> int: return self.N_ * self.M_
    def shape(self) -> Tuple[int, int]: return (self.N_, self.M_)
    def buffer(self) -> int: return self.t.data_ptr()
    def shared_buffer(self): return self.t

    def init(self, gain=None) -> None:
        """seed-99 uniform init, reference src/matrix.hpp:539-545 (host RNG, then upload)"""
        if isinstance(gain, (list, tuple, np.ndarray)):   # init(std::vector<r_t>) overload :547-549
            self.t.copy_(_torch().from_numpy(np.asarray(gain, dtype=np.float32).reshape(self.N_, self.M_)))
            _torch().cuda.current_stream(self.t.device).synchronize()
            return
        host = np.empty((self.N_, self.M_), dtype=np.float32)
        _lib.load().mggcn_init_uniform_host(host.ctypes.data, self.N_, self.M_, -1.0 if gain is None else gain)
        self.t.copy_(_torch().from_numpy(host))
        _torch().cuda.current_stream(self.t.device).synchronize()

    def zero(self, ctx: context) -> None:
        ctx.lib.mggcn_memset_zero(self.buffer(), self.size() * 4, ctx.stream(0))

    def copy_to(self, ctx: context, other: "dn_matrix") -> None:
        ctx.lib.mggcn_memcpy_d2d(other.buffer(), self.buffer(), self.size() * 4, ctx.stream(0))

    def copy(self, ctx: context) -> "dn_matrix":
        clone = dn_matrix(self.N_, self.M_, dtype=np.float32 if self.t.dtype.is_floating_point else np.int32,
                          device=self.t.device)
        self.copy_to(ctx, clone)
        return clone


class csr_matrix:
    """CSR<unsigned, unsigned, float>, reference src/matrix.hpp:214-468: file reader
    (:224-234, the PIGO-CSR-v2 layout), ``normalize`` (:340-390), ``transpose``
    (:392-453), ``as_dn`` (:328-337).  Host arrays are the source of truth for the
    preprocessing; ``device()`` uploads once and caches."""

    def __init__(self, indptr, indices=None, data=None, M: Optional[int] = None):
        if isinstance(indptr, (str, os.PathLike)):
            path = os.fspath(indptr)
            if not path.endswith(".bin"):
                raise matrix_error("File type is not supported.")      # src/matrix.hpp:282
            try:
                indptr, indices, data, n, M = datasets.read_csr(path)
            except datasets.format_error as e:
                raise matrix_error(str(e))
        self.indptr = np.ascontiguousarray(indptr, dtype=np.uint32)
        self.indices = np.ascontiguousarray(indices, dtype=np.uint32)
        self.data = np.ascontiguousarray(data, dtype=np.float32)
        self.N_ = int(self.indptr.shape[0] - 1)
        self.M_ = int(M if M is not None else self.N_)
        assert self.indices.shape[0] == self.nnz() and self.data.shape[0] == self.nnz()
        self._dev = None
        self._version = 0

    def n(self) -> int: return self.N_
    def m(self) -> int: return self.M_
    def nnz(self) -> int: return int(self.indptr[self.N_]) - int(self.indptr[0])
    def shape(self) -> Tuple[int, int]: return (self.N_, self.M_)
    def begin(self, i: int) -> int: return int(self.indptr[i])
    def end(self, i: int) -> int: return int(self.indptr[i + 1])

    def buffer(self):
        return self.indptr, self.indices, self.data

    def normalize(self, axis: bool = False) -> None:
        _lib.load().mggcn_csr_normalize_host(self.N_, self.M_, self.indptr.ctypes.data,
                                             self.indices.ctypes.data, self.data.ctypes.data, int(bool(axis)))
        self.invalidate()

    def invalidate(self) -> None:
        """Call after editing ``indptr`` / ``indices`` / ``data`` in place: drops the device copy AND the
        SpMM plans cached on this matrix (ops.spmm_plan_for) -- a sweep plan carries its own copy of the
        values, so a stale plan would silently multiply with the old matrix."""
        self._dev = None
        self._version = getattr(self, "_version", 0) + 1
        self.__dict__.pop("_spmm_plans", None)

    def transpose(self) -> "csr_matrix":
        nnz = self.nnz()
        t_indptr = np.empty(self.M_ + 1, dtype=np.uint32)
        t_indices = np.empty(nnz, dtype=np.uint32)
        t_data = np.empty(nnz, dtype=np.float32)
        _lib.load().mggcn_csr_transpose_host(self.N_, self.M_, self.indptr.ctypes.data,
                                             self.indices.ctypes.data, self.data.ctypes.data,
                                             t_indptr.ctypes.data, t_indices.ctypes.data, t_data.ctypes.data)
        return csr_matrix(t_indptr, t_indices, t_data, self.N_)

    def as_dn(self) -> np.ndarray:
        """dense copy on the HOST (test helper in the reference too), last duplicate wins"""
        out = np.zeros((self.N_, self.M_), dtype=np.float32)
        for v in range(self.N_):
            for e in range(self.begin(v), self.end(v)):
                out[v, self.indices[e]] = self.data[e]
        return out

    def device(self, device=None):
        """(indptr, indices, data) as device tensors (uint32 bits stored as int32)."""
        if self._dev is None:
            torch = _torch()
            dev = device or "cuda"
            self._dev = (torch.from_numpy(self.indptr.view(np.int32)).to(dev),
                         torch.from_numpy(self.indices.view(np.int32)).to(dev),
                         torch.from_numpy(self.data).to(dev))
            torch.cuda.current_stream(self._dev[0].device).synchronize()
        return self._dev
