"""oracle.py -- Python face of the CPU restatement.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg import this module; it is the checker, never the thing measured or shipped.

Heavy arithmetic (SpMM, GEMM, normalise, transpose, block split, weight init)
is the C/C++ restatement in ``mggcn_oracle.c`` / ``mggcn_oracle_init.cpp``
reached through ctypes; layer-level glue is numpy fp32.  Citations are relative
to ``/root/reference``.

Parity pin: the reference's own known-answer tests (test/test_gcn.cpp:98-249,
test/test_matrix.cpp:11-109) and the toyA/toyB fixtures -- see
``tests/test_oracle_kat.py``.  The reference's arithmetic lives in cuSPARSE /
cuBLAS (CUDA 11.4 era, un-pinned) which cannot run here, so nothing beyond those
vectors pins the fp32 summation order; the north-star tolerance (1e-4 relative)
absorbs it.
"""
from __future__ import annotations

import ctypes
import hashlib
import os
import subprocess
from typing import List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

u32p = ctypes.POINTER(ctypes.c_uint32)
i32p = ctypes.POINTER(ctypes.c_int32)
f32p = ctypes.POINTER(ctypes.c_float)


def _cpu_key() -> str:
    flags = ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    flags = line
                    break
    except OSError:
        pass
    return hashlib.sha1(flags.encode()).hexdigest()[:10]


def lib_path() -> str:
    return os.path.join(_HERE, "_build", f"liboracle-{_cpu_key()}.so")


def build(force: bool = False) -> str:
    """Compile the C restatement for THIS host's CPU (gcc -O3 -march=native -fopenmp)."""
    out = lib_path()
    srcs = [os.path.join(_HERE, s) for s in ("mggcn_oracle.c", "mggcn_oracle_init.cpp", "Makefile")]
    stale = force or not os.path.exists(out) or any(
        os.path.getmtime(s) > os.path.getmtime(out) for s in srcs)
    if stale:
        subprocess.check_call(
            ["make", "-s", "-C", _HERE, f"OUT={os.path.relpath(out, _HERE)}", "ARCH=-march=native"])
    return out


def _ptr(a: np.ndarray, ty):
    return a.ctypes.data_as(ty)


def lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(build())
        sz = ctypes.c_size_t
        L.orc_spmm_csr_f32.argtypes = [ctypes.c_uint32, u32p, u32p, f32p, f32p, sz, f32p, sz,
                                       ctypes.c_uint32, ctypes.c_float, ctypes.c_float]
        L.orc_spmm_csr_f64acc.argtypes = L.orc_spmm_csr_f32.argtypes
        L.orc_csr_normalize.argtypes = [ctypes.c_uint32, ctypes.c_uint32, u32p, u32p, f32p, ctypes.c_int]
        L.orc_csr_transpose.argtypes = [ctypes.c_uint32, ctypes.c_uint32, u32p, u32p, f32p, u32p, u32p, f32p]
        L.orc_csr_as_dn.argtypes = [ctypes.c_uint32, ctypes.c_uint32, u32p, u32p, f32p, f32p]
        L.orc_block_split_count.argtypes = [u32p, u32p, ctypes.c_uint32, ctypes.c_uint32, u32p,
                                            ctypes.c_uint32, u32p]
        L.orc_block_split_fill.argtypes = [u32p, u32p, f32p, ctypes.c_uint32, ctypes.c_uint32, u32p,
                                           ctypes.c_uint32, u32p, ctypes.POINTER(u32p),
                                           ctypes.POINTER(f32p)]
        g = [ctypes.c_int, ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32,
             ctypes.c_float, f32p, sz, f32p, sz, ctypes.c_float, f32p, sz]
        L.orc_gemm_f32.argtypes = g
        L.orc_gemm_f64acc.argtypes = g
        L.orc_init_uniform.argtypes = [f32p, sz, sz, ctypes.c_float]
        L.orc_default_gain_w.restype = ctypes.c_float
        L.orc_default_gain_b.restype = ctypes.c_float
        L.orc_num_threads.restype = ctypes.c_int
        L.orc_set_num_threads.argtypes = [ctypes.c_int]
        L.orc_leaky_relu_forward.argtypes = [f32p, f32p, sz, ctypes.c_float]
        L.orc_leaky_relu_backward.argtypes = [f32p, f32p, f32p, sz, ctypes.c_float]
        L.orc_broadcast_rows.argtypes = [f32p, f32p, sz, sz, ctypes.c_int]
        L.orc_scale_rows.argtypes = [f32p, f32p, sz, sz]
        L.orc_max_rows.argtypes = [f32p, f32p, sz, sz]
        L.orc_max_row_indices.argtypes = [f32p, i32p, sz, sz]
        L.orc_index_log_rows.argtypes = [f32p, i32p, f32p, sz, sz]
        L.orc_add_indexed_rows.argtypes = [f32p, i32p, ctypes.c_float, sz, sz]
        L.orc_is_equal.argtypes = [i32p, i32p, f32p, sz]
        L.orc_subtract_rows_exp.argtypes = [f32p, f32p, f32p, sz, sz]
        L.orc_axpby.argtypes = [f32p, f32p, ctypes.c_float, ctypes.c_float, sz]
        L.orc_aaxpby.argtypes = [f32p, f32p, ctypes.c_float, ctypes.c_float, sz]
        L.orc_adam_final.argtypes = [f32p, f32p, f32p, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                     ctypes.c_float, sz]
        L.orc_axpy.argtypes = [f32p, f32p, ctypes.c_float, sz]
        L.orc_scale_mat.argtypes = [f32p, ctypes.c_float, sz]
        L.orc_abssum.argtypes = [f32p, sz]
        L.orc_abssum.restype = ctypes.c_float
        _LIB = L
    return _LIB


def num_threads() -> int:
    return lib().orc_num_threads()


# ----------------------------------------------------------------------------
# CSR container (host) -- mirrors csr_matrix<u32,u32,f32>, src/matrix.hpp:214-221
# ----------------------------------------------------------------------------
class Csr:
    def __init__(self, indptr, indices, data, m: int):
        self.indptr = np.ascontiguousarray(indptr, dtype=np.uint32)
        self.indices = np.ascontiguousarray(indices, dtype=np.uint32)
        self.data = np.ascontiguousarray(data, dtype=np.float32)
        self.n = int(self.indptr.shape[0] - 1)
        self.m = int(m)

    @property
    def nnz(self) -> int:
        return int(self.indptr[-1] - self.indptr[0])

    def copy(self) -> "Csr":
        return Csr(self.indptr.copy(), self.indices.copy(), self.data.copy(), self.m)


def spmm(A: Csr, B: np.ndarray, C: Optional[np.ndarray] = None, alpha: float = 1.0,
         beta: float = 0.0, f64acc: bool = False) -> np.ndarray:
    """C = alpha*A*B + beta*C -- src/cuda_utils.hpp:15-32."""
    B = np.ascontiguousarray(B, dtype=np.float32)
    assert B.ndim == 2 and B.shape[0] == A.m, (B.shape, A.m)
    d = B.shape[1]
    if C is None:
        assert beta == 0.0
        C = np.empty((A.n, d), dtype=np.float32)
    assert C.dtype == np.float32 and C.shape == (A.n, d) and C.flags.c_contiguous
    fn = lib().orc_spmm_csr_f64acc if f64acc else lib().orc_spmm_csr_f32
    fn(A.n, _ptr(A.indptr, u32p), _ptr(A.indices, u32p), _ptr(A.data, f32p), _ptr(B, f32p), d,
       _ptr(C, f32p), d, d, alpha, beta)
    return C


def normalize(A: Csr, axis: bool = False) -> None:
    """In place -- src/matrix.hpp:340-390."""
    lib().orc_csr_normalize(A.n, A.m, _ptr(A.indptr, u32p), _ptr(A.indices, u32p),
                            _ptr(A.data, f32p), int(bool(axis)))


def transpose(A: Csr) -> Csr:
    """src/matrix.hpp:392-453 (serial order)."""
    t_indptr = np.empty(A.m + 1, dtype=np.uint32)
    t_indices = np.empty(max(A.nnz, 1), dtype=np.uint32)[:A.nnz]
    t_data = np.empty(max(A.nnz, 1), dtype=np.float32)[:A.nnz]
    lib().orc_csr_transpose(A.n, A.m, _ptr(A.indptr, u32p), _ptr(A.indices, u32p),
                            _ptr(A.data, f32p), _ptr(t_indptr, u32p), _ptr(t_indices, u32p),
                            _ptr(t_data, f32p))
    return Csr(t_indptr, t_indices, t_data, A.n)


def as_dn(A: Csr) -> np.ndarray:
    out = np.empty((A.n, A.m), dtype=np.float32)
    lib().orc_csr_as_dn(A.n, A.m, _ptr(A.indptr, u32p), _ptr(A.indices, u32p), _ptr(A.data, f32p),
                        _ptr(out, f32p))
    return out


def block_split(A: Csr, p: Sequence[int], q: Sequence[int]) -> List[List[Csr]]:
    """dist_row_csr_matrix ctor -- src/dist_matrix.hpp:215-259.  Returns blocks[i][j]."""
    p = [int(x) for x in p]
    qa = np.ascontiguousarray(q, dtype=np.uint32)
    nq = len(q) - 1
    out: List[List[Csr]] = []
    for i in range(len(p) - 1):
        rows = p[i + 1] - p[i]
        bip = np.empty((nq, rows + 1), dtype=np.uint32)
        lib().orc_block_split_count(_ptr(A.indptr, u32p), _ptr(A.indices, u32p), p[i], p[i + 1],
                                    _ptr(qa, u32p), nq, _ptr(bip, u32p))
        idx = [np.empty(max(int(bip[j, rows]), 1), dtype=np.uint32) for j in range(nq)]
        dat = [np.empty(max(int(bip[j, rows]), 1), dtype=np.float32) for j in range(nq)]
        ip = (u32p * nq)(*[_ptr(a, u32p) for a in idx])
        dp = (f32p * nq)(*[_ptr(a, f32p) for a in dat])
        lib().orc_block_split_fill(_ptr(A.indptr, u32p), _ptr(A.indices, u32p), _ptr(A.data, f32p),
                                   p[i], p[i + 1], _ptr(qa, u32p), nq, _ptr(bip, u32p), ip, dp)
        out.append([Csr(bip[j].copy(), idx[j][:int(bip[j, rows])], dat[j][:int(bip[j, rows])],
                        int(qa[j + 1] - qa[j])) for j in range(nq)])
    return out


def gemm(A: np.ndarray, B: np.ndarray, C: Optional[np.ndarray] = None, alpha: float = 1.0,
         beta: float = 0.0, A_T: bool = False, B_T: bool = False, f64acc: bool = False) -> np.ndarray:
    """C = alpha*op(A)*op(B) + beta*C, row-major -- src/cuda_utils.hpp:149-172."""
    A = np.ascontiguousarray(A, dtype=np.float32)
    B = np.ascontiguousarray(B, dtype=np.float32)
    M, K = (A.shape[1], A.shape[0]) if A_T else A.shape
    K2, N = (B.shape[1], B.shape[0]) if B_T else B.shape
    assert K == K2, (A.shape, B.shape, A_T, B_T)
    if C is None:
        assert beta == 0.0
        C = np.empty((M, N), dtype=np.float32)
    assert C.shape == (M, N) and C.dtype == np.float32 and C.flags.c_contiguous
    fn = lib().orc_gemm_f64acc if f64acc else lib().orc_gemm_f32
    fn(int(A_T), int(B_T), M, N, K, alpha, _ptr(A, f32p), A.shape[1], _ptr(B, f32p), B.shape[1], beta,
       _ptr(C, f32p), N)
    return C


def init_uniform(n_rows: int, n_cols: int, gain: Optional[float] = None) -> np.ndarray:
    """dn_matrix::init -- src/matrix.hpp:539-545 (seed 99, libstdc++)."""
    if gain is None:
        gain = lib().orc_default_gain_w()
    out = np.empty((n_rows, n_cols), dtype=np.float32)
    lib().orc_init_uniform(_ptr(out, f32p), n_rows, n_cols, gain)
    return out


def gain_b() -> float:
    return lib().orc_default_gain_b()


# ----------------------------------------------------------------------------
# Element-wise / row kernels (thin wrappers over the C restatement)
# ----------------------------------------------------------------------------
def leaky_relu_forward(x: np.ndarray, alpha: float = 0.01) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty_like(x)
    lib().orc_leaky_relu_forward(_ptr(x, f32p), _ptr(out, f32p), x.size, alpha)
    return out


def leaky_relu_backward(act: np.ndarray, G: np.ndarray, alpha: float = 0.01) -> np.ndarray:
    act = np.ascontiguousarray(act, dtype=np.float32)
    G = np.ascontiguousarray(G, dtype=np.float32)
    out = np.empty_like(G)
    lib().orc_leaky_relu_backward(_ptr(act, f32p), _ptr(G, f32p), _ptr(out, f32p), G.size, alpha)
    return out


def softmax_rows(H: np.ndarray, f64acc: bool = False) -> np.ndarray:
    """softmax::operator() -- src/gcn.hpp:651-675: max_rows, subtract_rows_exp,
    row sums by a GEMM with a ones vector, scale_rows."""
    H = np.ascontiguousarray(H, dtype=np.float32)
    n, m = H.shape
    maxs = np.empty(n, dtype=np.float32)
    lib().orc_max_rows(_ptr(H, f32p), _ptr(maxs, f32p), H.size, m)
    E = np.empty_like(H)
    lib().orc_subtract_rows_exp(_ptr(H, f32p), _ptr(maxs, f32p), _ptr(E, f32p), H.size, m)
    R = gemm(E, np.ones((m, 1), dtype=np.float32), f64acc=f64acc)
    lib().orc_scale_rows(_ptr(E, f32p), _ptr(R, f32p), E.size, m)
    return E


def softmax_cross_entropy(H: np.ndarray, Y: np.ndarray, n_global: Optional[int] = None,
                          f64acc: bool = False) -> Tuple[float, float, np.ndarray, np.ndarray]:
    """softmax_cross_entropy_loss::operator() -- src/gcn.hpp:785-818.
    Returns (sum|log p_y|, sum[y==argmax], G = (O - onehot)/n_global, O).
    The caller divides the two sums by n (src/gcn.hpp:817 / :929)."""
    Y = np.ascontiguousarray(Y, dtype=np.int32).reshape(-1)
    O = softmax_rows(H, f64acc)
    n, m = O.shape
    if n_global is None:
        n_global = n
    P = np.empty(n, dtype=np.int32)
    lib().orc_max_row_indices(_ptr(O, f32p), _ptr(P, i32p), O.size, m)
    L = np.empty(n, dtype=np.float32)
    lib().orc_index_log_rows(_ptr(O, f32p), _ptr(Y, i32p), _ptr(L, f32p), O.size, m)
    G = O.copy()
    lib().orc_add_indexed_rows(_ptr(G, f32p), _ptr(Y, i32p), -1.0, G.size, m)
    lib().orc_scale_mat(_ptr(G, f32p), np.float32(1.0) / np.float32(n_global), G.size)
    T = np.empty(n, dtype=np.float32)
    lib().orc_is_equal(_ptr(Y, i32p), _ptr(P, i32p), _ptr(T, f32p), n)
    loss_sum = lib().orc_abssum(_ptr(L, f32p), n)
    acc_sum = lib().orc_abssum(_ptr(T, f32p), n)
    return float(loss_sum), float(acc_sum), G, O


# ----------------------------------------------------------------------------
# Layers -- numpy glue over the kernels above
# ----------------------------------------------------------------------------
class Linear:
    """linear<r_t> -- src/gcn.hpp:88-189."""

    def __init__(self, n_in: int, n_out: int, backward_out: bool = True, f64acc: bool = False):
        self.f64acc = f64acc                               # exact-accumulation twin (see Gcn)
        self.W = init_uniform(n_in, n_out)                 # gcn.hpp:108, matrix.hpp:539
        self.b = init_uniform(1, n_out, gain_b())          # gcn.hpp:109
        self.G_W = np.zeros_like(self.W)
        self.G_b = np.zeros_like(self.b)
        self.backward_out = backward_out
        self.X = None
        self.step = 0
        self.mW = self.vW = self.mb = self.vb = None

    def forward(self, X: np.ndarray) -> np.ndarray:        # gcn.hpp:116-123
        XW = np.empty((X.shape[0], self.W.shape[1]), dtype=np.float32)
        lib().orc_broadcast_rows(_ptr(self.b, f32p), _ptr(XW, f32p), XW.size, XW.shape[1], 1)
        gemm(X, self.W, XW, 1.0, 1.0, f64acc=self.f64acc)
        self.X = X
        return XW

    def backward(self, G: np.ndarray) -> Optional[np.ndarray]:   # gcn.hpp:125-139
        ones = np.ones((1, G.shape[0]), dtype=np.float32)
        self.G_b = gemm(ones, G, f64acc=self.f64acc)
        self.G_W = gemm(self.X, G, A_T=True, f64acc=self.f64acc)
        return gemm(G, self.W, B_T=True, f64acc=self.f64acc) if self.backward_out else None

    def adam_update(self, lr, beta1, beta2, weight_decay, eps):  # gcn.hpp:146-172
        L = lib()
        if self.mW is None:
            self.mW = np.zeros_like(self.W); self.vW = np.zeros_like(self.W)
            self.mb = np.zeros_like(self.b); self.vb = np.zeros_like(self.b)
            self.step = 0
        self.step += 1
        bc1 = np.float32(1 - beta1 ** self.step)
        bc2 = np.float32(1 - beta2 ** self.step)
        L.orc_axpy(_ptr(self.W, f32p), _ptr(self.G_W, f32p), weight_decay, self.W.size)
        L.orc_axpby(_ptr(self.G_W, f32p), _ptr(self.mW, f32p), 1 - beta1, beta1, self.W.size)
        L.orc_axpby(_ptr(self.G_b, f32p), _ptr(self.mb, f32p), 1 - beta1, beta1, self.b.size)
        L.orc_aaxpby(_ptr(self.G_W, f32p), _ptr(self.vW, f32p), 1 - beta2, beta2, self.W.size)
        L.orc_aaxpby(_ptr(self.G_b, f32p), _ptr(self.vb, f32p), 1 - beta2, beta2, self.b.size)
        L.orc_adam_final(_ptr(self.W, f32p), _ptr(self.mW, f32p), _ptr(self.vW, f32p), lr, bc1, bc2,
                         eps, self.W.size)
        L.orc_adam_final(_ptr(self.b, f32p), _ptr(self.mb, f32p), _ptr(self.vb, f32p), lr, bc1, bc2,
                         eps, self.b.size)


class GcnLayer:
    """gcn_layer -- src/gcn.hpp:411-518.  ``spmm_fwd``/``spmm_bwd`` are callables
    B -> A*B so the single-GPU and the P-shard simulations share this class."""

    def __init__(self, spmm_fwd, spmm_bwd, n_in: int, n_out: int, activation: bool,
                 backward_spmm: bool = True, f64acc: bool = False, residual_layer: bool = False):
        self.spmm_fwd, self.spmm_bwd = spmm_fwd, spmm_bwd
        self.lin = Linear(n_in, n_out, backward_spmm, f64acc)   # gcn.hpp:430 (backward_out = backward_spmm)
        # gcn.hpp:430: res_lin(in == out || !residual_layer ? nullopt : linear(name, in, out, backward_spmm))
        self.residual_layer = residual_layer
        self.res_lin = Linear(n_in, n_out, backward_spmm, f64acc) if residual_layer and n_in != n_out else None
        self.gemm_first = n_out <= n_in                    # HW.m()==AHW.m(), gcn.hpp:439
        self.activation = activation
        self.backward_spmm = backward_spmm

    def linears(self):
        return [self.lin] + ([self.res_lin] if self.res_lin is not None else [])

    def forward(self, H: np.ndarray) -> np.ndarray:        # gcn.hpp:437-458
        self.H = H
        if self.gemm_first:
            HW = self.lin.forward(H)
            Z = self.spmm_fwd(HW)
        else:
            HW = self.spmm_fwd(H)
            Z = self.lin.forward(HW)
        if self.activation:
            Z = leaky_relu_forward(Z)
        if self.res_lin is not None:                        # gcn.hpp:453-454: res_lin(ctx, H, AHW, discard = false)
            R = self.res_lin
            lib().orc_broadcast_rows(_ptr(R.b, f32p), _ptr(Z, f32p), Z.size, Z.shape[1], 0)      # AHW += b   (gcn.hpp:117)
            gemm(H, R.W, Z, 1.0, 1.0, f64acc=R.f64acc)                                            # AHW += H.W (gcn.hpp:120)
            R.X = H
        elif self.residual_layer:                           # gcn.hpp:455-456: axpy(ctx, H, AHW, 1)
            lib().orc_axpy(_ptr(np.ascontiguousarray(H), f32p), _ptr(Z, f32p), 1.0, Z.size)
        self.AHW = Z
        return Z

    def backward(self, G: np.ndarray) -> Optional[np.ndarray]:   # gcn.hpp:460-489
        # the sign source of leaky_relu_backward is AHW AFTER the residual add (gcn.hpp:464 reads the buffer
        # the forward pass left, :453-456) -- restated as the reference does it
        T = leaky_relu_backward(self.AHW, G) if self.activation else G
        if self.gemm_first:
            G_HW = self.spmm_bwd(T) if self.backward_spmm else T
            G_out = self.lin.backward(G_HW)
        else:
            self.lin.X = self.H                             # lin.setX(H)
            G_HW = self.lin.backward(T)
            G_out = None if G_HW is None else (self.spmm_bwd(G_HW) if self.backward_spmm else G_HW)
        if self.res_lin is not None:                        # gcn.hpp:484-485: res_lin->backward(ctx, G, G_out, false)
            R = self.res_lin
            ones = np.ones((1, G.shape[0]), dtype=np.float32)
            R.G_b = gemm(ones, G, f64acc=R.f64acc)
            R.G_W = gemm(R.X, G, A_T=True, f64acc=R.f64acc)
            if R.backward_out and G_out is not None:
                gemm(G, R.W, G_out, 1.0, 1.0, B_T=True, f64acc=R.f64acc)
        elif self.residual_layer and G_out is not None:     # gcn.hpp:486-487: axpy(ctx, G, G_out, 1)
            lib().orc_axpy(_ptr(np.ascontiguousarray(G), f32p), _ptr(G_out, f32p), 1.0, G_out.size)
        return G_out


class Gcn:
    """gcn -- src/gcn.hpp:937-995.  A is normalised by column, A_T = A^T, layers
    get (A_T, A): forward multiplies by A_T, backward by A (gcn.hpp:946-955)."""

    def __init__(self, A: Csr, sizes: Sequence[int], f64acc: bool = False, residual_layer: bool = False):
        """``f64acc``: every SpMM / GEMM sum is accumulated in fp64 and rounded to fp32 once (inputs,
        outputs and all element-wise math stay fp32).  The reference's sums run in cuSPARSE / cuBLAS in
        an unspecified fp32 order; over K = n = 233 k terms (G_b = 1^T G, G_W = X^T G at the Reddit shape)
        a sequential fp32 sum is itself ~1e-4 away from the exact one, so full-size parity is judged
        against this twin (same algorithm, order-free), with the fp32 restatement's own distance to it
        reported next to the device's."""
        A = A.copy()
        normalize(A, True)
        A_T = transpose(A)
        self.A_fwd, self.A_bwd = A_T, A
        self.f64acc = f64acc
        self.layers = [GcnLayer(lambda B, M=A_T: spmm(M, B, f64acc=f64acc), lambda B, M=A: spmm(M, B, f64acc=f64acc),
                                sizes[i - 1], sizes[i], i + 1 < len(sizes), i != 1, f64acc, residual_layer)
                       for i in range(1, len(sizes))]

    def forward(self, H: np.ndarray) -> np.ndarray:
        for layer in self.layers:
            H = layer.forward(H)
        return H

    def train_forward(self, X: np.ndarray, Y: np.ndarray) -> Tuple[float, float]:
        H = self.forward(np.ascontiguousarray(X, dtype=np.float32))
        ls, ac, self.G, self.O = softmax_cross_entropy(H, Y, f64acc=self.f64acc)
        n = np.float32(H.shape[0])
        return float(np.float32(ls) / n), float(np.float32(ac) / n)

    def backward(self) -> None:
        G = self.G
        for layer in reversed(self.layers):
            G = layer.backward(G)

    def adam_update(self, lr=1e-2, beta1=0.9, beta2=0.999, weight_decay=5e-4, eps=1e-8) -> None:
        for layer in self.layers:
            for lin in layer.linears():                     # gcn.hpp:497-500
                lin.adam_update(lr, beta1, beta2, weight_decay, eps)


# ----------------------------------------------------------------------------
# Distributed restatement: P shards simulated in one process
# ----------------------------------------------------------------------------
def dist_spmm(blocks: List[List[Csr]], B_shards: List[np.ndarray],
              C_shards: Optional[List[np.ndarray]] = None, alpha: float = 1.0, beta: float = 0.0
              ) -> List[np.ndarray]:
    """matmul(dist_context, dist_row_csr, dist_row_dn, ...) -- src/cuda_utils.hpp:47-92:
    C_j = beta*C_j + alpha * sum_i A[j,i] * B_i, rounds i = 0..P-1 in order, round 0
    with the caller's beta and the rest with beta = 1."""
    P = len(blocks)
    d = B_shards[0].shape[1]
    if C_shards is None:
        assert beta == 0.0
        C_shards = [np.empty((blocks[j][0].n, d), dtype=np.float32) for j in range(P)]
    for i in range(P):
        for j in range(P):
            spmm(blocks[j][i], B_shards[i], C_shards[j], alpha, beta if i == 0 else 1.0)
    return C_shards


class DistGcn:
    """dist_gcn<true,...> on P simulated GPUs -- src/gcn.hpp:997-1056 with
    src/main.cpp:134-153: classes padded to a multiple of P, 1D row partition
    p[i] = i*n/P, W/b replicated (same seed-99 init on every rank),
    G_W / G_b summed over ranks (all-reduce), loss scaled by the GLOBAL n."""

    def __init__(self, A: Csr, sizes: Sequence[int], P: int, residual_layer: bool = False):
        sizes = list(sizes)
        sizes[-1] = (sizes[-1] + P - 1) // P * P               # main.cpp:135
        assert A.n % P == 0                                      # dist_matrix.hpp:428
        self.P, self.n = P, A.n
        self.p = [i * A.n // P for i in range(P + 1)]           # main.cpp:139-141
        A = A.copy()
        normalize(A, True)                                       # main.cpp:143
        A_T = transpose(A)                                       # main.cpp:144
        self.Ad = block_split(A, self.p, self.p)                 # main.cpp:148
        self.A_Td = block_split(A_T, self.p, self.p)             # main.cpp:149
        self.sizes = sizes
        # layers get (A_T, A): forward uses A_Td, backward Ad (gcn.hpp:1023)
        self.ranks = []
        for _ in range(P):
            self.ranks.append([GcnLayer(None, None, sizes[i - 1], sizes[i], i + 1 < len(sizes), i != 1,
                                        residual_layer=residual_layer)
                               for i in range(1, len(sizes))])

    def _shard(self, X: np.ndarray) -> List[np.ndarray]:
        return [np.ascontiguousarray(X[self.p[j]:self.p[j + 1]]) for j in range(self.P)]

    def train_forward(self, X: np.ndarray, Y: np.ndarray) -> Tuple[float, float]:
        P = self.P
        H = self._shard(np.ascontiguousarray(X, dtype=np.float32))
        Ys = self._shard(np.ascontiguousarray(Y, dtype=np.int32).reshape(-1, 1))
        nl = len(self.ranks[0])
        for li in range(nl):
            L = [self.ranks[j][li] for j in range(P)]
            for j in range(P):
                L[j].H = H[j]
            if L[0].gemm_first:
                HW = [L[j].lin.forward(H[j]) for j in range(P)]
                Z = dist_spmm(self.A_Td, HW)
            else:
                HW = dist_spmm(self.A_Td, H)
                Z = [L[j].lin.forward(HW[j]) for j in range(P)]
            if L[0].activation:
                Z = [leaky_relu_forward(z) for z in Z]
            for j in range(P):                                   # residual branch, row-local (gcn.hpp:572-575)
                if L[j].res_lin is not None:
                    R = L[j].res_lin
                    lib().orc_broadcast_rows(_ptr(R.b, f32p), _ptr(Z[j], f32p), Z[j].size, Z[j].shape[1], 0)
                    gemm(H[j], R.W, Z[j], 1.0, 1.0)
                    R.X = H[j]
                elif L[j].residual_layer:
                    lib().orc_axpy(_ptr(np.ascontiguousarray(H[j]), f32p), _ptr(Z[j], f32p), 1.0, Z[j].size)
            for j in range(P):
                L[j].AHW = Z[j]
            H = Z
        ls = ac = np.float32(0)
        self.G = []
        for j in range(P):
            l, a, G, _ = softmax_cross_entropy(H[j], Ys[j], n_global=self.n)   # gcn.hpp:908
            ls += np.float32(l); ac += np.float32(a)
            self.G.append(G)
        n = np.float32(self.n)
        return float(ls / n), float(ac / n)

    def backward(self) -> None:
        P = self.P
        G = self.G
        nl = len(self.ranks[0])
        for li in reversed(range(nl)):
            L = [self.ranks[j][li] for j in range(P)]
            G_in = G
            T = [leaky_relu_backward(L[j].AHW, G[j]) if L[0].activation else G[j] for j in range(P)]
            if L[0].gemm_first:
                G_HW = dist_spmm(self.Ad, T) if L[0].backward_spmm else T
                G = [L[j].lin.backward(G_HW[j]) for j in range(P)]
            else:
                for j in range(P):
                    L[j].lin.X = L[j].H
                G_HW = [L[j].lin.backward(T[j]) for j in range(P)]
                G = dist_spmm(self.Ad, G_HW) if L[0].backward_spmm else G_HW
            Gin = G_in
            for j in range(P):                                   # residual branch (gcn.hpp:603-606)
                if L[j].res_lin is not None:
                    R = L[j].res_lin
                    R.G_b = gemm(np.ones((1, Gin[j].shape[0]), dtype=np.float32), Gin[j])
                    R.G_W = gemm(R.X, Gin[j], A_T=True)
                    if R.backward_out and G[j] is not None:
                        gemm(Gin[j], R.W, G[j], 1.0, 1.0, B_T=True)
                elif L[j].residual_layer and G[j] is not None:
                    lib().orc_axpy(_ptr(np.ascontiguousarray(Gin[j]), f32p), _ptr(G[j], f32p), 1.0, G[j].size)
            # all-reduce(sum) of G_W and G_b over ranks -- gcn.hpp:236-240, cuda_utils.hpp:304-313
            for pick in ((lambda l: l.lin), (lambda l: l.res_lin)):
                if pick(L[0]) is None:
                    continue
                GW = pick(L[0]).G_W.copy(); Gb = pick(L[0]).G_b.copy()
                for j in range(1, P):
                    GW += pick(L[j]).G_W; Gb += pick(L[j]).G_b
                for j in range(P):
                    pick(L[j]).G_W = GW.copy(); pick(L[j]).G_b = Gb.copy()

    def adam_update(self, lr=1e-2, beta1=0.9, beta2=0.999, weight_decay=5e-4, eps=1e-8) -> None:
        for layers in self.ranks:
            for layer in layers:
                for lin in layer.linears():
                    lin.adam_update(lr, beta1, beta2, weight_decay, eps)
