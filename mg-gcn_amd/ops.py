"""The reference's operator overload set (src/cuda_utils.hpp) over the C ABI.

Same function names, argument order and meaning as the reference's free
functions taking ``(context, matrix objects...)``; each one enqueues exactly one
C-ABI call on ``ctx.cuda_streams[0]`` (the reference's compute stream).
Shape preconditions the reference only ``assert``s raise ``ValueError`` here.
"""
from __future__ import annotations

import os

from typing import Optional

from .matrix import context, csr_matrix, dn_matrix


def _req(cond: bool, what: str) -> None:
    if not cond:
        raise ValueError(what)


class spmm_buffer:
    """What get_matmul_buffer returns: the reference's opaque cuSPARSE workspace
    (src/cuda_utils.hpp:94-102) becomes the row-split plan of the HIP kernel."""

    def __init__(self, lib, handle: int):
        self.lib, self.handle = lib, handle
        self.max_d = 0
        self.d_hint = 0
        self.version = 0            # csr_matrix._version the plan was built from

    def num_items(self) -> int: return self.lib.mggcn_spmm_plan_num_items(self.handle)
    def num_split_rows(self) -> int: return self.lib.mggcn_spmm_plan_num_split_rows(self.handle)
    def num_sweep_tasks(self) -> int: return self.lib.mggcn_spmm_plan_num_sweep_tasks(self.handle)
    def num_launches(self, d: int) -> int: return self.lib.mggcn_spmm_plan_num_launches(self.handle, int(d))
    def nbytes(self) -> int: return self.lib.mggcn_spmm_plan_bytes(self.handle)

    def describe(self) -> str:
        """what the plan builder measured and decided (mggcn_spmm_plan_describe)"""
        import ctypes
        buf = ctypes.create_string_buffer(4096)
        self.lib.mggcn_spmm_plan_describe(self.handle, buf, 4096)
        return buf.value.decode()

    def __del__(self):
        try:
            if self.handle:
                self.lib.mggcn_spmm_plan_destroy(self.handle)
                self.handle = 0
        except Exception:
            pass


def get_matmul_buffer(ctx: context, A: csr_matrix, B: dn_matrix, C: dn_matrix, alpha: float = 1.0,
                      beta: float = 0.0, max_d: Optional[int] = None) -> spmm_buffer:
    """reference src/cuda_utils.hpp:94-102"""
    _req(A.m() == B.n(), "A.m() != B.n()")
    _req(A.n() == C.n() and B.m() == C.m(), "C shape mismatch")
    return spmm_plan_for(ctx, A, max(int(max_d or 0), B.m()), B.m())


def spmm_plan_for(ctx: context, A: csr_matrix, max_d: int, d_hint: int) -> spmm_buffer:
    """One plan per (matrix, device, form): the layers of a model multiply by the same two matrices,
    so the plan (0.9 GB and ~1 s of host work on the Reddit shape) is built once and shared.  The
    form depends on the width only through narrow (<= 64: lanes per row) vs wide."""
    form = ("narrow", (int(d_hint) + 15) // 16) if 1 <= int(d_hint) <= 64 else ("wide", 0)
    knobs = tuple(sorted((k, v) for k, v in os.environ.items() if k.startswith("MGGCN_SPMM_")))   # tuning / tests
    key = (ctx.rank, form, knobs)
    cache = A.__dict__.setdefault("_spmm_plans", {})
    buf = cache.get(key)
    if buf is None or buf.max_d < max_d:
        ctx.set()
        h = ctx.lib.mggcn_spmm_plan_create_for(A.n(), A.m(), A.indptr.ctypes.data, A.indices.ctypes.data,
                                               A.data.ctypes.data, int(max_d), int(d_hint))
        buf = cache[key] = spmm_buffer(ctx.lib, h)
        buf.max_d, buf.d_hint, buf.version = int(max_d), int(d_hint), A._version
    return buf


def prebuild_plans(ctx: context, wants, max_parallel: int = 4) -> None:
    """wants = [(csr_matrix, max_d, d_hint), ...]: the plans a model is going to ask for, built SIDE BY SIDE (up to
    four host threads; ctypes releases the GIL and every plan builder threads its own passes) and filed in the
    matrices' caches, instead of one by one inside the first epoch: a single-GPU model multiplies by two matrices at
    two widths -- four plans of 0.4-1.0 s of host work each at the Reddit shape, 2.9 s in a row."""
    from concurrent.futures import ThreadPoolExecutor
    knobs = tuple(sorted((k, v) for k, v in os.environ.items() if k.startswith("MGGCN_SPMM_")))
    jobs = {}
    for A, max_d, d_hint in wants:
        form = ("narrow", (int(d_hint) + 15) // 16) if 1 <= int(d_hint) <= 64 else ("wide", 0)
        key = (ctx.rank, form, knobs)
        have = A.__dict__.setdefault("_spmm_plans", {}).get(key)
        if have is not None and have.max_d >= max_d:
            continue
        jk = (id(A), key)
        if jk not in jobs or jobs[jk][1] < max_d:
            jobs[jk] = (A, int(max_d), int(d_hint), key)
    if not jobs:
        return

    def build(job):
        A, max_d, d_hint, _ = job
        ctx.lib.mggcn_set_device(ctx.rank)
        return ctx.lib.mggcn_spmm_plan_create_for(A.n(), A.m(), A.indptr.ctypes.data, A.indices.ctypes.data,
                                                  A.data.ctypes.data, max_d, d_hint)
    todo = list(jobs.values())
    workers = max(1, min(max_parallel, len(todo)))
    ctx.lib.mggcn_spmm_plan_concurrent_builders(workers)        # every builder threads over its share of the cores
    try:
        with ThreadPoolExecutor(max_workers=workers) as pool:
            handles = list(pool.map(build, todo))
    finally:
        ctx.lib.mggcn_spmm_plan_concurrent_builders(1)
    for (A, max_d, d_hint, key), h in zip(todo, handles):
        buf = A.__dict__["_spmm_plans"][key] = spmm_buffer(ctx.lib, h)
        buf.max_d, buf.d_hint, buf.version = max_d, d_hint, A._version


def matmul(ctx: context, A, B: dn_matrix, C: dn_matrix, *args, **kw) -> None:
    """Overloads, as in the reference:
       matmul(ctx, csr A, B, C, ext_buffer, alpha, beta)        src/cuda_utils.hpp:27-32
       matmul(ctx, dn  A, B, C, alpha, beta, A_T=False, B_T=False)  src/cuda_utils.hpp:158-172"""
    if isinstance(A, csr_matrix):
        return _spmm(ctx, A, B, C, *args, **kw)
    return _gemm(ctx, A, B, C, *args, **kw)


def _spmm(ctx: context, A: csr_matrix, B: dn_matrix, C: dn_matrix, ext_buffer: Optional[spmm_buffer],
          alpha: float, beta: float, flags: int = 0, slope: float = 0.01, stream_id: int = 0) -> None:
    _req(A.m() == B.n() and B.m() == C.m() and A.n() == C.n(), "SpMM shape mismatch")
    ctx.set()
    if ext_buffer is not None and ext_buffer.version != A._version:
        # the matrix was edited in place after the plan was built (csr_matrix.normalize / invalidate): the
        # reference's cuSPARSE workspace holds no values, so the same call sequence must keep working
        ext_buffer = spmm_plan_for(ctx, A, ext_buffer.max_d, ext_buffer.d_hint or B.m())
    ip, ix, dv = A.device(ctx.device)
    ctx.lib.mggcn_spmm_csr_f32(ctx.stream(stream_id), ext_buffer.handle if ext_buffer else None, A.n(), A.m(),
                               ip.data_ptr(), ix.data_ptr(), dv.data_ptr(), B.buffer(), B.m(), C.buffer(),
                               C.m(), B.m(), alpha, beta, flags, slope)


def _gemm(ctx: context, A: dn_matrix, B: dn_matrix, C: dn_matrix, alpha: float, beta: float,
          A_T: bool = False, B_T: bool = False) -> None:
    A_n, A_m, B_n, B_m = A.n(), A.m(), B.n(), B.m()
    if A_T:
        A_n, A_m = A_m, A_n
    if B_T:
        B_n, B_m = B_m, B_n
    _req(A_m == B_n, "GEMM inner dimensions differ")
    _req(A_n == C.n() and B_m == C.m(), "GEMM output shape mismatch")
    ctx.set()
    ws_bytes = ctx.lib.mggcn_gemm_workspace_bytes(int(A_T), int(B_T), A_n, B_m, A_m)
    ws = ctx.workspace(ws_bytes)
    ctx.lib.mggcn_gemm_f32(ctx.stream(0), int(A_T), int(B_T), A_n, B_m, A_m, alpha, A.buffer(), A.m(),
                           B.buffer(), B.m(), beta, C.buffer(), C.m(), ws.data_ptr() if ws is not None else None,
                           ws_bytes)


def linear_forward(ctx: context, X: dn_matrix, W: dn_matrix, b: dn_matrix, XW: dn_matrix) -> None:
    """XW = X.W + 1 b^T in one GEMM pass (bias in the epilogue) -- the fused form of the
    reference's broadcast_rows + sgemm(beta = 1), src/gcn.hpp:116-123."""
    _req(X.m() == W.n() and XW.n() == X.n() and XW.m() == W.m() and b.m() == W.m() and b.n() == 1, "linear shape")
    ctx.set()
    ws_bytes = ctx.lib.mggcn_gemm_workspace_bytes(0, 0, X.n(), W.m(), X.m())
    ws = ctx.workspace(ws_bytes)
    ctx.lib.mggcn_gemm_bias_f32(ctx.stream(0), 0, 0, X.n(), W.m(), X.m(), 1.0, X.buffer(), X.m(), W.buffer(), W.m(),
                                b.buffer(), XW.buffer(), XW.m(), ws.data_ptr() if ws is not None else None, ws_bytes)


def matmul_lrelu_backward(ctx: context, A: dn_matrix, B: dn_matrix, Z: dn_matrix, C: dn_matrix, alpha: float = 1.0,
                          A_T: bool = False, B_T: bool = False, slope: float = 0.01) -> None:
    """C = (alpha op(A) op(B)) .* (Z > 0 ? 1 : slope): the GEMM that produces a layer's input gradient with the
    leaky_relu_backward of the layer below folded into its epilogue (src/gcn.hpp:135-137 + :462-468)."""
    A_n, A_m, B_n, B_m = A.n(), A.m(), B.n(), B.m()
    if A_T:
        A_n, A_m = A_m, A_n
    if B_T:
        B_n, B_m = B_m, B_n
    _req(A_m == B_n and A_n == C.n() and B_m == C.m() and Z.shape() == C.shape(), "GEMM + mask shape mismatch")
    ctx.set()
    ws_bytes = ctx.lib.mggcn_gemm_workspace_bytes(int(A_T), int(B_T), A_n, B_m, A_m)
    ws = ctx.workspace(ws_bytes)
    ctx.lib.mggcn_gemm_lrelu_bwd_f32(ctx.stream(0), int(A_T), int(B_T), A_n, B_m, A_m, alpha, A.buffer(), A.m(),
                                     B.buffer(), B.m(), Z.buffer(), Z.m(), slope, C.buffer(), C.m(),
                                     ws.data_ptr() if ws is not None else None, ws_bytes)


def linear_backward_weights(ctx: context, X: dn_matrix, G: dn_matrix, G_W: dn_matrix, G_b: dn_matrix) -> None:
    """G_W = X^T G and G_b = 1^T G in ONE pass over G (the fused form of the two sgemms of linear::backward,
    src/gcn.hpp:125-134): the column sums ride on the B tiles of the X^T G kernel."""
    _req(X.n() == G.n() and G_W.shape() == (X.m(), G.m()) and G_b.shape() == (1, G.m()), "linear backward shape")
    ctx.set()
    ws_bytes = ctx.lib.mggcn_gemm_tn_colsum_workspace_bytes(X.m(), G.m(), X.n())
    ws = ctx.workspace(ws_bytes)
    ctx.lib.mggcn_gemm_tn_colsum_f32(ctx.stream(0), X.m(), G.m(), X.n(), 1.0, X.buffer(), X.m(), G.buffer(), G.m(),
                                     G_W.buffer(), G_W.m(), G_b.buffer(), ws.data_ptr() if ws is not None else None, ws_bytes)


def gather_rows(ctx: context, src: dn_matrix, indices, dst: dn_matrix, stream_id: int = 0) -> None:
    """dst[k, :] = src[indices[k], :] (halo pack; indices: device uint32 tensor viewed as int32 storage)"""
    n_idx = int(indices.numel())
    _req(dst.n() >= n_idx and dst.m() == src.m(), "gather_rows shape")
    ctx.lib.mggcn_gather_rows_f32(ctx.stream(stream_id), src.buffer(), src.m(), indices.data_ptr(), n_idx, src.m(),
                                  dst.buffer(), dst.m())


# ---- element-wise / row kernels: src/cuda_utils.hpp:470-748 wrappers -----------------
def leaky_relu_forward(ctx: context, in_: dn_matrix, out: dn_matrix, alpha: float = 0.01) -> None:
    _req(in_.shape() == out.shape(), "shape mismatch")
    ctx.lib.mggcn_leaky_relu_forward_f32(ctx.stream(0), in_.buffer(), out.buffer(), in_.size(), alpha)


def leaky_relu_backward(ctx: context, in_: dn_matrix, G_in: dn_matrix, G_out: dn_matrix,
                        alpha: float = 0.01) -> None:
    _req(in_.shape() == G_in.shape() == G_out.shape(), "shape mismatch")
    ctx.lib.mggcn_leaky_relu_backward_f32(ctx.stream(0), in_.buffer(), G_in.buffer(), G_out.buffer(),
                                          in_.size(), alpha)


def broadcast_rows(ctx: context, row: dn_matrix, mat: dn_matrix, discard: bool = True) -> None:
    _req(row.m() == mat.m(), "row width mismatch")
    ctx.lib.mggcn_broadcast_rows_f32(ctx.stream(0), row.buffer(), mat.buffer(), mat.size(), mat.m(), int(discard))


def scale_rows(ctx: context, mat: dn_matrix, scalar: dn_matrix) -> None:
    _req(mat.n() == scalar.n(), "row count mismatch")
    ctx.lib.mggcn_scale_rows_f32(ctx.stream(0), mat.buffer(), scalar.buffer(), mat.size(), mat.m())


def max_rows(ctx: context, mat: dn_matrix, maxs: dn_matrix) -> None:
    _req(mat.n() == maxs.n() and maxs.m() == 1, "maxs must be n x 1")
    ctx.lib.mggcn_max_rows_f32(ctx.stream(0), mat.buffer(), maxs.buffer(), mat.size(), mat.m())


def max_row_indices(ctx: context, mat: dn_matrix, maxs: dn_matrix) -> None:
    _req(mat.n() == maxs.n() and maxs.m() == 1, "maxs must be n x 1")
    ctx.lib.mggcn_max_row_indices_f32(ctx.stream(0), mat.buffer(), maxs.buffer(), mat.size(), mat.m())


def index_log_rows(ctx: context, mat: dn_matrix, indices: dn_matrix, values: dn_matrix) -> None:
    _req(mat.n() == indices.n() and indices.m() == 1 and values.n() == mat.n() and values.m() == 1, "shape")
    ctx.lib.mggcn_index_log_rows_f32(ctx.stream(0), mat.buffer(), indices.buffer(), values.buffer(), mat.size(),
                                     mat.m())


def add_indexed_rows(ctx: context, mat: dn_matrix, indices: dn_matrix, alpha: float) -> None:
    _req(mat.n() == indices.n() and indices.m() == 1, "shape")
    ctx.lib.mggcn_add_indexed_rows_f32(ctx.stream(0), mat.buffer(), indices.buffer(), alpha, mat.size(), mat.m())


def is_equal(ctx: context, mat1: dn_matrix, mat2: dn_matrix, out: dn_matrix) -> None:
    _req(mat1.shape() == mat2.shape() == out.shape(), "shape mismatch")
    ctx.lib.mggcn_is_equal_i32(ctx.stream(0), mat1.buffer(), mat2.buffer(), out.buffer(), mat1.size())


def subtract_rows_exp(ctx: context, mat: dn_matrix, scalar: dn_matrix, out: dn_matrix) -> None:
    _req(mat.n() == scalar.n() and scalar.m() == 1 and mat.shape() == out.shape(), "shape")
    ctx.lib.mggcn_subtract_rows_exp_f32(ctx.stream(0), mat.buffer(), scalar.buffer(), out.buffer(), mat.size(),
                                        mat.m())


def axpy(ctx: context, A: dn_matrix, B: dn_matrix, alpha: float) -> None:
    _req(A.shape() == B.shape(), "shape mismatch")
    ctx.lib.mggcn_axpy_f32(ctx.stream(0), A.buffer(), B.buffer(), alpha, A.size())


def axpby(ctx: context, A: dn_matrix, B: dn_matrix, alpha: float, beta: float) -> None:
    _req(A.shape() == B.shape(), "shape mismatch")
    ctx.lib.mggcn_axpby_f32(ctx.stream(0), A.buffer(), B.buffer(), alpha, beta, A.size())


def aaxpby(ctx: context, A: dn_matrix, B: dn_matrix, alpha: float, beta: float) -> None:
    _req(A.shape() == B.shape(), "shape mismatch")
    ctx.lib.mggcn_aaxpby_f32(ctx.stream(0), A.buffer(), B.buffer(), alpha, beta, A.size())


def adam_final(ctx: context, param: dn_matrix, m: dn_matrix, v: dn_matrix, lr: float, c1: float, c2: float,
               eps: float) -> None:
    _req(param.shape() == m.shape() == v.shape(), "shape mismatch")
    ctx.lib.mggcn_adam_final_f32(ctx.stream(0), param.buffer(), m.buffer(), v.buffer(), lr, c1, c2, eps,
                                 param.size())


def scale_mat(ctx: context, mat: dn_matrix, scalar: float) -> None:
    ctx.lib.mggcn_scale_mat_f32(ctx.stream(0), mat.buffer(), scalar, mat.size())


def abssum(ctx: context, A: dn_matrix, result_device) -> None:
    """cublasSasum (src/cuda_utils.hpp:362-371).  ``result_device``: 1-element float32
    device tensor; enqueue-only (the reference's call blocks the host)."""
    ctx.lib.mggcn_abssum_f32(ctx.stream(0), A.buffer(), A.size(), result_device.data_ptr())


# ---- fused tail ------------------------------------------------
def softmax_xent_fused(ctx: context, H: dn_matrix, Y: dn_matrix, grad_scale: float, sums_device,
                       out: Optional[dn_matrix] = None) -> None:
    """softmax + argmax + log-prob + gradient in one pass; in place on H, or H -> out (the loss layer's copy = True:
    the pass is the copy)"""
    _req(H.n() == Y.n() and Y.m() == 1, "labels must be n x 1")
    if out is None:
        out = H
    _req(out.n() == H.n() and out.m() == H.m(), "fused loss: gradient matrix must have the logits' shape")
    ctx.lib.mggcn_softmax_xent_fused_from_f32(ctx.stream(0), H.buffer(), out.buffer(), Y.buffer(), H.n(), H.m(),
                                              grad_scale, sums_device.data_ptr())


def adam_fused(ctx: context, param: dn_matrix, grad: dn_matrix, m: dn_matrix, v: dn_matrix, lr: float,
               beta1: float, beta2: float, weight_decay: float, c1: float, c2: float, eps: float) -> None:
    _req(param.shape() == grad.shape() == m.shape() == v.shape(), "shape mismatch")
    ctx.lib.mggcn_adam_fused_f32(ctx.stream(0), param.buffer(), grad.buffer(), m.buffer(), v.buffer(), lr,
                                 beta1, beta2, weight_decay, c1, c2, eps, param.size())


class adam_table:
    """Device table of every (param, grad, m, v) quadruple of a model for mggcn_adam_multi_f32: built once
    (the buffers of a model never move), one launch per epoch instead of two per layer."""

    def __init__(self, ctx: context, tensors) -> None:
        """tensors: [(param, grad, m, v, weight_decay_on)] of dn_matrix"""
        import numpy as np
        torch = __import__("torch")
        dt = np.dtype([("param", "<u8"), ("grad", "<u8"), ("m", "<u8"), ("v", "<u8"), ("size", "<u8"),
                       ("wd", "<f4"), ("first_block", "<u4")])
        assert dt.itemsize == 48
        tab = np.zeros(len(tensors), dtype=dt)
        blocks = 0
        for k, (p, g, m, v, wd) in enumerate(tensors):
            _req(p.shape() == g.shape() == m.shape() == v.shape(), "shape mismatch")
            tab[k] = (p.buffer(), g.buffer(), m.buffer(), v.buffer(), p.size(), wd, blocks)
            blocks += ctx.lib.mggcn_adam_multi_blocks(p.size())
        self.n, self.blocks = len(tensors), blocks
        self.keep = tensors                                       # the table holds raw pointers
        self.dev = torch.from_numpy(tab.view(np.uint8).copy()).to(ctx.device)
        torch.cuda.current_stream(self.dev.device).synchronize()

    def step(self, ctx: context, lr: float, beta1: float, beta2: float, c1: float, c2: float, eps: float) -> None:
        ctx.lib.mggcn_adam_multi_f32(ctx.stream(0), self.dev.data_ptr(), self.n, self.blocks, lr, beta1, beta2, c1, c2, eps)
