"""1D row (vertex) partition of the GCN: one process per GPU over torch.distributed.

Reference: the single-process, P-GPU classes ``dist_context`` (src/dist_matrix.hpp:12-90),
``dist_row_csr_matrix`` (:170-260), ``dist_row_dn_matrix`` (:394-532), ``repl_dn_matrix``
(:534-639), ``dist_sparse_linear`` / ``dist_row_linear`` / ``dist_gcn_layer`` /
``dist_row_softmax_cross_entropy_loss`` / ``dist_gcn`` (src/gcn.hpp:50-86, :191-296,
:520-637, :872-935, :997-1056) and the pipelined distributed SpMM
(src/cuda_utils.hpp:57-92).  Same names, same math, same timer names -- but each
process holds ONLY its own GPU's pieces (rank j owns row block j of A, shard j of
every activation / gradient, a replica of W): `M[i]` of the reference becomes the
local `M.local`.  The single-process, all-GPUs-in-one-thread form lives in the C++
header layer (mg-gcn_amd/host).

Exchange step (the reference broadcasts every shard in P rounds, double-buffered,
src/cuda_utils.hpp:61-89).  Two modes:

``allgather`` (default, MI355X-first): ONE all-gather of the [n/P x d] shards into a
    resident [n x d] buffer on the high-priority comm stream -- on the xGMI full mesh
    every GPU pushes to all 7 peers at once -- while the compute stream already
    multiplies the local diagonal block A[j,j] (needs no communication); the
    remaining P-1 blocks are pre-merged at partition time into one CSR with global
    column indices and run over the gathered buffer (beta = 1).  For P > 2 the exchange
    is cut into K pieces of every shard (K all-gathers queued back to back, the remote
    block cut by piece with renumbered columns): the SpMM over piece c overlaps the
    transfer of piece c+1.
    C_j = A[j,j].B_j + sum_{i!=j} A[j,i].B_i  -- the reference's sum, regrouped.
``halo``: the same split, but every rank sends each peer only the rows of its shard that the
    peer's blocks reference (index lists built at partition time, packed by a gather kernel,
    exchanged with ONE all-to-all of variable-size pieces, remote block renumbered to the
    receive layout).  On a graph where every rank touches every row (the random synthetic
    Reddit) this moves what the all-gather moves; on a partitioned sparse graph it moves the
    boundary only.
``rounds``: the reference's schedule, one broadcast + one block SpMM per round i,
    double-buffered, accumulating in round order (used for order-exact parity tests
    and as the `-S`-style baseline).
"""
from __future__ import annotations

import os

import ctypes
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _lib, ops
from .gcn import MGGCN_SPMM_LEAKY_RELU, _SQRT_1_3, adam_update_all, link_fused_backward, softmax_cross_entropy_loss
from .matrix import context, csr_matrix, dn_matrix


def _torch():
    import torch
    return torch


def _dist():
    import torch.distributed as dist
    return dist


# --------------------------------------------------------------------------------------
# host-side partitioning (pure numpy + the C ABI's host functions; no GPU needed)
# --------------------------------------------------------------------------------------
def partition_bounds(n: int, P: int) -> List[int]:
    """p[i] = i*n/P (reference src/main.cpp:139-141); requires n % P == 0
    (src/dist_matrix.hpp:428; the data prep pads n to a multiple of 8)."""
    if n % P != 0:
        raise ValueError(f"n = {n} is not a multiple of the number of GPUs {P}")
    return [i * n // P for i in range(P + 1)]


def split_row_block(A: csr_matrix, row_begin: int, row_end: int, q: Sequence[int]) -> List[csr_matrix]:
    """Row block [row_begin,row_end) of A cut at the column boundaries q into len(q)-1
    CSR blocks with block-local column indices: one row of the reference's P x P grid
    (src/dist_matrix.hpp:215-259)."""
    lib = _lib.load()
    qa = np.ascontiguousarray(q, dtype=np.uint32)
    nq = len(q) - 1
    rows = row_end - row_begin
    bip = np.empty((nq, rows + 1), dtype=np.uint32)
    lib.mggcn_csr_block_split_count_host(A.indptr.ctypes.data, A.indices.ctypes.data, row_begin, row_end,
                                         qa.ctypes.data, nq, bip.ctypes.data)
    idx = [np.empty(int(bip[j, rows]), dtype=np.uint32) for j in range(nq)]
    dat = [np.empty(int(bip[j, rows]), dtype=np.float32) for j in range(nq)]
    ip = (ctypes.c_void_p * nq)(*[a.ctypes.data for a in idx])
    dp = (ctypes.c_void_p * nq)(*[a.ctypes.data for a in dat])
    lib.mggcn_csr_block_split_fill_host(A.indptr.ctypes.data, A.indices.ctypes.data, A.data.ctypes.data,
                                        row_begin, row_end, qa.ctypes.data, nq, bip.ctypes.data, ip, dp)
    return [csr_matrix(bip[j].copy(), idx[j], dat[j], int(qa[j + 1] - qa[j])) for j in range(nq)]


def split_local_remote(A: csr_matrix, row_begin: int, row_end: int, col_begin: Optional[int] = None,
                       col_end: Optional[int] = None) -> Tuple[csr_matrix, csr_matrix]:
    """The all-gather form of the same row block: (diagonal block with local column
    indices, everything else with GLOBAL column indices).  Built with the same block
    splitter: cut at [0, row_begin, row_end, n] and merge the two outer blocks.  ``col_begin/col_end``:
    the diagonal column range when A is already a ROW BLOCK (rows 0..rows of A = global rows col_begin..)."""
    n = A.m()
    cb, ce = (row_begin, row_end) if col_begin is None else (col_begin, col_end)
    q = [0, cb, ce, n]
    left, diag, right = split_row_block(A, row_begin, row_end, q)
    row_end_g = ce
    rows = row_end - row_begin
    ln = np.diff(left.indptr.astype(np.int64))
    rn = np.diff(right.indptr.astype(np.int64))
    indptr = np.zeros(rows + 1, dtype=np.int64)
    np.cumsum(ln + rn, out=indptr[1:])
    nnz = int(indptr[-1])
    indices = np.empty(nnz, dtype=np.uint32)
    data = np.empty(nnz, dtype=np.float32)
    # per row: left entries (global index = local) then right entries (+ row_end)
    lpos = (np.repeat(indptr[:-1], ln) + (np.arange(int(ln.sum())) - np.repeat(left.indptr[:-1].astype(np.int64), ln)))
    rpos = (np.repeat(indptr[:-1] + ln, rn) + (np.arange(int(rn.sum())) - np.repeat(right.indptr[:-1].astype(np.int64), rn)))
    indices[lpos] = left.indices
    data[lpos] = left.data
    indices[rpos] = right.indices + np.uint32(row_end_g)
    data[rpos] = right.data
    return diag, csr_matrix(indptr.astype(np.uint32), indices, data, n)


def default_chunks(P: int) -> int:
    """Pieces the exchange of one SpMM is cut into (all-gather mode).  With one piece only the
    diagonal block (1/P of the work) overlaps the exchange; with K pieces the SpMM over piece c
    runs while piece c+1 is still on the wire.  Two pieces at P = 2 (one xGMI link between the pair:
    60 MB per exchange is of the order of the diagonal block's 0.7 ms, so half the remote block should
    start early), four from P = 3 on.  A piece is not free: rank 0's whole epoch at P = 8 with the exchange switched off is
    2.96 / 3.04 / 3.08 / 3.25 / 3.57 ms with 1 / 2 / 3 / 4 / 6 pieces (launches, partial-sum combines, one more pass over C
    each); with the exchange modelled as a delay at 7 x 50 GB/s it is 4.54 / 3.74 / 3.53 / 3.54 / 3.69 ms, at 7 x 36 GB/s
    5.23 / 4.32 / 4.10 / 4.03 / 4.05 -- three or four pieces at P = 8, four at P = 4 (3 links), two or three at P = 2 (one link):
    profiles/experiments/rank_epoch_model_r04.log.  Override with MGGCN_DIST_CHUNKS."""
    import os
    env = os.environ.get("MGGCN_DIST_CHUNKS")
    if env:
        return max(1, int(env))
    return 1 if P <= 1 else (2 if P == 2 else 4)


def chunk_bounds(rows: int, K: int) -> List[int]:
    return [c * rows // K for c in range(K + 1)]


def split_remote_chunks(remote: csr_matrix, P: int, rows: int, K: int) -> List[csr_matrix]:
    """Cuts the merged remote block (global column g = s*rows + i: row i of rank s's shard) into
    K matrices by the PIECE of the shard the column lives in, i in [cb[c], cb[c+1]), and renumbers
    the columns to the layout one all-gather of that piece produces: s*len_c + (i - cb[c]).
    sum_c remote_c . gathered_c == remote . gathered  (same products, regrouped)."""
    cb = np.asarray(chunk_bounds(rows, K), dtype=np.int64)
    if K == 1:
        return [remote]
    ip = remote.indptr.astype(np.int64)
    g = remote.indices.astype(np.int64)
    src, i = g // rows, g % rows
    cid = np.searchsorted(cb, i, side="right") - 1
    row_ids = np.repeat(np.arange(remote.n(), dtype=np.int64), np.diff(ip))
    counts = np.bincount(cid * remote.n() + row_ids, minlength=K * remote.n()).reshape(K, remote.n())
    order = np.argsort(cid, kind="stable")                 # grouped by piece, CSR order kept inside
    new_col = (src * (cb[cid + 1] - cb[cid]) + (i - cb[cid])).astype(np.uint32)
    out, at = [], 0
    for c in range(K):
        nz = int(counts[c].sum())
        sel = order[at:at + nz]
        at += nz
        indptr = np.zeros(remote.n() + 1, dtype=np.int64)
        np.cumsum(counts[c], out=indptr[1:])
        out.append(csr_matrix(indptr.astype(np.uint32), new_col[sel], remote.data[sel], int(P * (cb[c + 1] - cb[c]))))
    return out


def halo_need_lists(blocks: List[csr_matrix], r: int) -> List[np.ndarray]:
    """need[s] = sorted distinct rows of rank s's shard that the blocks A[{r, s}] of rank r reference
    (block-local column indices); empty for s == r."""
    return [np.unique(b.indices).astype(np.uint32) if s != r else np.empty(0, dtype=np.uint32)
            for s, b in enumerate(blocks)]


def halo_volume_matrix(A: csr_matrix, P: int) -> np.ndarray:
    """V[i, j] = distinct rows of shard j that rank i's row block references (i != j): the
    communication volume in rows per exchange of mode="halo", the matrix the reference's data-prep
    script prints for a partition (test/data/prep.py:237-244).  The all-gather form moves
    (P - 1) * n / P rows into every rank whatever the graph."""
    p = partition_bounds(A.n(), P)
    V = np.zeros((P, P), dtype=np.int64)
    for i in range(P):
        blocks = split_row_block(A, p[i], p[i + 1], p)
        for j, x in enumerate(halo_need_lists(blocks, i)):
            V[i, j] = len(x)
    return V


def merge_blocks_halo(blocks: List[csr_matrix], need: List[np.ndarray], r: int) -> csr_matrix:
    """The remote blocks of one row block merged into one CSR whose columns index the RECEIVE
    buffer of the halo exchange: pieces in source-rank order, piece s holding need[s] in order.
    Row order and in-row order (source rank, then original order) are kept."""
    rows = blocks[0].n()
    off = np.zeros(len(blocks) + 1, dtype=np.int64)
    np.cumsum([len(x) for x in need], out=off[1:])
    lens = [np.diff(b.indptr.astype(np.int64)) if s != r else np.zeros(rows, dtype=np.int64) for s, b in enumerate(blocks)]
    indptr = np.zeros(rows + 1, dtype=np.int64)
    np.cumsum(np.sum(lens, axis=0), out=indptr[1:])
    nnz = int(indptr[-1])
    indices = np.empty(nnz, dtype=np.uint32)
    data = np.empty(nnz, dtype=np.float32)
    start = indptr[:-1].copy()
    for s, b in enumerate(blocks):
        if s == r or b.nnz() == 0:
            continue
        ln = lens[s]
        pos = np.repeat(start, ln) + (np.arange(int(ln.sum())) - np.repeat(b.indptr[:-1].astype(np.int64), ln))
        indices[pos] = (off[s] + np.searchsorted(need[s], b.indices)).astype(np.uint32)
        data[pos] = b.data
        start = start + ln
    return csr_matrix(indptr.astype(np.uint32), indices, data, int(off[-1]))


# --------------------------------------------------------------------------------------
# host-staged collectives (gloo): CPU rehearsal of the exchange step and the fallback when
# several ranks share one GPU.  Pure torch.distributed on CPU tensors -- no GPU needed.
# --------------------------------------------------------------------------------------
def gloo_all_gather_rows(host_shard, P: int, group=None):
    """[rows x d] per rank -> [P*rows x d] on every rank, rank order (== ncclAllGather /
    the P broadcasts of reference src/dist_matrix.hpp:458-467)."""
    torch, dist = _torch(), _dist()
    # ONE output tensor: gloo's list form of all_gather (P output tensors + a concatenation) is 4x slower at these sizes
    # -- 28 ms against 6.5 ms for four ranks x [2912 x 128] on loopback, the same as P broadcasts
    # (profiles/experiments/gloo_allgather_r04.log).  That, not plan building, was the 45.9 ms against 12.4 / 9.8 ms of
    # the all-gather schedule in the round-3 rehearsals (28 exchanges per epoch); the RCCL path never took this branch.
    host_shard = host_shard.contiguous()
    out = torch.empty((P * host_shard.shape[0],) + tuple(host_shard.shape[1:]), dtype=host_shard.dtype)
    dist.all_gather_into_tensor(out, host_shard, group=group)
    return out


def gloo_broadcast_rows(host_shard, shape, dtype, root: int, rank: int, group=None):
    """one round of the reference's schedule: shard `root` to everybody"""
    torch, dist = _torch(), _dist()
    buf = host_shard.contiguous() if rank == root else torch.empty(tuple(shape), dtype=dtype)
    src = dist.get_global_rank(group, root) if group is not None else root
    dist.broadcast(buf, src=src, group=group)
    return buf


def gloo_all_reduce_sum(host_flat, group=None):
    _dist().all_reduce(host_flat, group=group)
    return host_flat


# --------------------------------------------------------------------------------------
class host_comm:
    """Rank / world size + the host-array collectives of the one-off partition step (rank-local load, halo
    lists).  Needs no GPU with a gloo group (the CPU tests); dist_context is one of these plus the device side."""

    def __init__(self, group=None, device=None):
        dist = _dist()
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed must be initialised (one process per GPU)")
        self.group = group
        self.P = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.backend = dist.get_backend(group)
        self._comm_device = device                       # nccl moves host arrays through this device

    def _nccl(self) -> bool:
        return self.backend == "nccl"

    def _to_comm(self, arr: np.ndarray):
        t = _torch().from_numpy(np.ascontiguousarray(arr))
        return t.to(self._comm_device) if self._nccl() else t

    def host_all_reduce(self, arr: np.ndarray, op: str = "sum") -> np.ndarray:
        dist = _dist()
        t = self._to_comm(arr)
        dist.all_reduce(t, op=dist.ReduceOp.SUM if op == "sum" else dist.ReduceOp.MAX, group=self.group)
        return t.cpu().numpy()

    def host_all_to_all(self, pieces: Sequence[np.ndarray]) -> List[np.ndarray]:
        """pieces[s] (2-D, same trailing shape and dtype) goes to rank s; returns what every rank sent here,
        in rank order.  Row counts are exchanged first."""
        torch, dist = _torch(), _dist()
        P = self.P
        width = pieces[0].shape[1:]
        dt = pieces[0].dtype
        counts = np.array([int(x.shape[0]) for x in pieces], dtype=np.int64)
        cin = self._to_comm(counts)
        cout = torch.empty_like(cin)
        dist.all_to_all_single(cout, cin, group=self.group)
        recv_rows = [int(x) for x in cout.cpu().numpy()]
        send = self._to_comm(np.concatenate([np.ascontiguousarray(x).reshape((-1,) + width) for x in pieces], axis=0))
        recv = torch.empty((sum(recv_rows),) + tuple(width), dtype=send.dtype, device=send.device)
        dist.all_to_all_single(recv, send, recv_rows, [int(c) for c in counts], group=self.group)
        out = recv.cpu().numpy().astype(dt, copy=False)
        offs = np.concatenate([[0], np.cumsum(recv_rows)])
        return [out[offs[k]:offs[k + 1]] for k in range(P)]


class dist_context(host_comm):
    """reference src/dist_matrix.hpp:12-90, one rank's view.  ``overlap`` selects the
    comm stream exactly like bcast_stream_id() (:20-22); ``-S`` on the CLI clears it."""

    def __init__(self, overlap: bool = True, device_index: Optional[int] = None, group=None):
        host_comm.__init__(self, group)
        self.overlap = overlap
        # bench.py's exchange pass: extra events around every exchange (comm-stream time of the collectives, time the
        # compute stream spends waiting for each piece).  Off in timed epochs: ~10 more event records per SpMM.
        self.profile_exchange = False
        # MGGCN_DIST_SELF_GATHER=1 (tests of the transport): run the all-gather with ONE rank too -- the only way to put
        # ProcessGroupNCCL's all_gather_into_tensor on a one-GPU box; by default a single rank exchanges nothing
        self.self_gather = os.environ.get("MGGCN_DIST_SELF_GATHER", "0") == "1"
        if device_index is None:
            device_index = self.rank % max(_lib.require_gpu(), 1)
        self.ctx = context(device_index)
        self._comm_device = self.ctx.device

    def size(self) -> int: return self.P
    def bcast_stream_id(self) -> int: return 1 if self.overlap else 0
    def sync(self) -> None: self.ctx.sync()
    def record(self, name, stream_id): self.ctx.record(name, stream_id)
    def wait(self, name, stream_id): self.ctx.wait(name, stream_id)
    def register_timer(self, name, beg, end): self.ctx.register_timer(name, beg, end)
    def measure(self, name): return self.ctx.measure(name)

    def dump_timers(self, out, prefix: str = "") -> None:
        self.ctx.dump_timers(out, f"{prefix}{self.rank}_")      # "<epoch>_<rank>_<name>:ms"

    # -- collectives ------------------------------------------------------------------
    def all_gather_rows(self, shard, out, stream_id: int):
        """out[rank*rows:(rank+1)*rows] = shard on every rank.  Returns a handle whose
        .wait(stream_id) orders the result before later work on that stream."""
        torch, dist = _torch(), _dist()
        if self._nccl():
            with torch.cuda.stream(self.ctx.cuda_streams[stream_id]):
                work = dist.all_gather_into_tensor(out, shard, group=self.group, async_op=True)
            return _Pending(self, work, None, None)
        # gloo (several ranks sharing one GPU): stage through the host
        self.ctx.cuda_streams[0].synchronize()
        self.ctx.cuda_streams[stream_id].synchronize()
        return _Pending(self, None, gloo_all_gather_rows(shard.detach().cpu(), self.P, self.group), out)

    def broadcast_rows(self, shard, out, root: int, stream_id: int):
        torch, dist = _torch(), _dist()
        src = dist.get_global_rank(self.group, root) if self.group is not None else root
        if self._nccl():
            with torch.cuda.stream(self.ctx.cuda_streams[stream_id]):
                if self.rank == root:
                    out.copy_(shard)
                work = dist.broadcast(out, src=src, group=self.group, async_op=True)
            return _Pending(self, work, None, None)
        self.ctx.cuda_streams[0].synchronize()
        self.ctx.cuda_streams[stream_id].synchronize()
        host = gloo_broadcast_rows(shard.detach().cpu() if self.rank == root else None, out.shape,
                                   out.dtype, root, self.rank, self.group)
        return _Pending(self, None, host, out)

    def all_reduce_sum(self, tensors: Sequence, stream_id: int = 0) -> None:
        """in-place sum over ranks of several small tensors as ONE flat buffer (the
        reference all-reduces G_W and G_b separately, src/gcn.hpp:236-240)."""
        torch, dist = _torch(), _dist()
        st = self.ctx.cuda_streams[stream_id]
        with torch.cuda.stream(st):
            flat = torch.cat([t.reshape(-1) for t in tensors])
            if self._nccl():
                dist.all_reduce(flat, group=self.group)
            else:
                st.synchronize()
                flat.copy_(gloo_all_reduce_sum(flat.cpu(), self.group))
            off = 0
            for t in tensors:
                t.copy_(flat[off:off + t.numel()].view_as(t))
                off += t.numel()


    def all_to_all_rows(self, send, recv, send_rows: Sequence[int], recv_rows: Sequence[int], stream_id: int):
        """Variable-size row exchange: rows [sum(send_rows[:s]), +send_rows[s]) of ``send`` go to rank s,
        ``recv`` receives recv_rows[s] rows from rank s, in rank order."""
        torch, dist = _torch(), _dist()
        if self._nccl():
            with torch.cuda.stream(self.ctx.cuda_streams[stream_id]):
                work = dist.all_to_all_single(recv, send, list(recv_rows), list(send_rows), group=self.group,
                                              async_op=True)
            return _Pending(self, work, None, None)
        self.ctx.cuda_streams[0].synchronize()
        self.ctx.cuda_streams[stream_id].synchronize()
        host_out = torch.empty(recv.shape, dtype=recv.dtype)
        dist.all_to_all_single(host_out, send.detach().cpu(), list(recv_rows), list(send_rows), group=self.group)
        return _Pending(self, None, host_out, recv)

    def all_reduce_sum_async(self, flat, after_stream_id: int = 0):
        """In-place sum over ranks of ONE flat tensor on the comm stream, ordered after the work
        already queued on ``after_stream_id``; returns a handle whose .wait(stream_id) orders the
        result before later work on that stream.  The compute stream keeps going meanwhile."""
        torch, dist = _torch(), _dist()
        cs = self.bcast_stream_id()
        if self._nccl():
            if cs != after_stream_id:
                self.ctx.record("__allreduce-ready", after_stream_id)
                self.ctx.wait("__allreduce-ready", cs)
            with torch.cuda.stream(self.ctx.cuda_streams[cs]):
                work = dist.all_reduce(flat, group=self.group, async_op=True)
            return _Pending(self, work, None, None)
        self.ctx.cuda_streams[after_stream_id].synchronize()
        return _Pending(self, None, gloo_all_reduce_sum(flat.detach().cpu(), self.group), flat)


class _Pending:
    def __init__(self, dctx, work, host, out):
        self.dctx, self.work, self.host, self.out = dctx, work, host, out

    def wait(self, stream_id: int) -> None:
        torch = _torch()
        with torch.cuda.stream(self.dctx.ctx.cuda_streams[stream_id]):
            if self.work is not None:
                self.work.wait()                 # stream-level dependency, host does not block
            else:
                self.out.copy_(self.host.to(self.out.device))


# --------------------------------------------------------------------------------------
class dist_row_dn_matrix:
    """reference src/dist_matrix.hpp:394-532: rows split evenly over the ranks; this
    process holds shard ``rank`` as ``.local`` ([n/P x m])."""

    def __init__(self, dctx: dist_context, N, M=None, buffer=None, dtype=np.float32):
        if isinstance(N, np.ndarray):                        # ctor from a host matrix (:440-447)
            full = N if N.ndim == 2 else N.reshape(-1, 1)
            p = partition_bounds(full.shape[0], dctx.P)
            self.N_ = full.shape[0]
            self.local = dn_matrix.from_numpy(full[p[dctx.rank]:p[dctx.rank + 1]], dctx.ctx.device)
            return
        if isinstance(N, tuple):
            N, M = N
        if N % dctx.P != 0:
            raise ValueError("N % P != 0")                   # assert, :428
        self.N_ = int(N)
        self.local = dn_matrix(N // dctx.P, M, buffer, dtype, dctx.ctx.device)

    def n(self) -> int: return self.N_
    def m(self) -> int: return self.local.m()
    def shape(self): return (self.N_, self.local.m())


class repl_dn_matrix:
    """reference src/dist_matrix.hpp:534-639: a full copy on every rank."""

    def __init__(self, dctx: dist_context, N, M=None):
        if isinstance(N, tuple):
            N, M = N
        self.local = dn_matrix(N, M, device=dctx.ctx.device)

    def n(self): return self.local.n()
    def m(self): return self.local.m()
    def shape(self): return self.local.shape()

    def init(self, dctx: dist_context, gain=None) -> None:
        """reference :601-609 initialises on GPU 0 and broadcasts; every rank running the
        same seed-99 host generator yields the same bits with no traffic."""
        self.local.init(gain)

    def zero(self, dctx: dist_context) -> None:
        self.local.zero(dctx.ctx)

    def allreduce(self, dctx: dist_context) -> None:          # :587-592
        dctx.all_reduce_sum([self.local.t])


class dist_row_csr_matrix:
    """reference src/dist_matrix.hpp:170-260: A cut into P x P blocks, block (i,j) =
    rows of rank i x rows-of-H of rank j, block-local column indices.  This process
    builds only ITS row of blocks, plus the (diagonal, merged-remote) pair used by the
    all-gather schedule."""

    def __init__(self, dctx: dist_context, A: csr_matrix, p: Sequence[int], q: Sequence[int],
                 chunks: Optional[int] = None, row_block: bool = False):
        """``row_block``: A holds ONLY this rank's rows p[r]..p[r+1] (rank-local load, load_rank_local) instead
        of the whole matrix; everything below needs no more than that."""
        assert list(p) == list(q), "the reference only ever passes p == q (src/main.cpp:148-149)"
        self.N_, self.M_ = int(p[-1]), A.m()
        self.p = list(p)
        r = dctx.rank
        rb, re = (0, p[r + 1] - p[r]) if row_block else (p[r], p[r + 1])
        assert not row_block or A.n() == re
        self.blocks = split_row_block(A, rb, re, q)                       # A[{r, j}]
        self.diag, self.remote = split_local_remote(A, rb, re, p[r], p[r + 1])
        rows = p[r + 1] - p[r]
        K = default_chunks(dctx.P) if chunks is None else int(chunks)
        K = max(1, min(K, rows))
        self.chunk_bounds = chunk_bounds(rows, K)
        self.remote_chunks = split_remote_chunks(self.remote, dctx.P, rows, K)
        self._dctx, self._rank, self._P = dctx, r, dctx.P
        self.halo = None

    def build_halo(self, device) -> "dist_row_csr_matrix":
        """Index lists and the renumbered remote block of the halo exchange (mode="halo").  need[s] = rows of
        shard s this rank's blocks reference (local knowledge); what the PEERS need of this rank's shard comes
        from ONE all-to-all of those lists -- no rank ever looks at another rank's row block."""
        if self.halo is not None:
            return self
        torch = _torch()
        r, P, p = self._rank, self._P, self.p
        need = halo_need_lists(self.blocks, r)
        if P > 1:
            got = self._dctx.host_all_to_all([x.astype(np.int64).reshape(-1, 1) for x in need])
            send = [g.reshape(-1).astype(np.uint32) for g in got]
        else:
            send = [np.empty(0, dtype=np.uint32)]
        send_idx = np.concatenate(send) if P > 1 else np.empty(0, dtype=np.uint32)
        self.halo = {
            "recv_rows": [int(len(x)) for x in need],
            "send_rows": [int(len(x)) for x in send],
            "send_idx": torch.from_numpy(send_idx.astype(np.int64)).to(torch.int32).to(device),   # uint32 bits
            "remote": merge_blocks_halo(self.blocks, need, r),
        }
        return self

    def n(self): return self.N_
    def m(self): return self.M_
    def __getitem__(self, ij): return self.blocks[ij[1]]


def load_rank_local_host(dctx: host_comm, dirname: str):
    """The dataset as ONE RANK needs it, read without ever holding the whole graph (the reference's single process
    loads everything once, src/main.cpp:82-85, and splits it for its P GPUs, :143-153; one process per GPU doing the
    same would hold P whole copies of A, A^T and all P x P blocks on one host -- papers100M: 8 x 40 GB):
      * rows p[r]..p[r+1] of graph.bin / features.bin / labels.bin through the indptr offsets (datasets.read_csr_rows),
      * column sums of A by ONE all-reduce of the per-rank partial sums (fp64) -> this rank's rows of the
        column-normalised A D^-1 (csr_matrix::normalize(true), src/matrix.hpp:340-364).  Exactness: with the unit
        (integer) edge weights the reference's data-prep writes (test/data/prep.py:113) the sums are exact in any
        order and the result is BITWISE the whole-graph path's (asserted in tests/test_dist_cpu.py); with
        non-unit weights the reference and mggcn_csr_normalize_host add in fp32 in row order while these partial
        sums are fp64, so values may differ from the P = 1 matrix in the last ulp (parity unpinned there),
      * this rank's rows of (A D^-1)^T by ONE all-to-all: every entry (i, c) goes to the owner of column c, which
        receives its transposed rows already in increasing source-row order (src/matrix.hpp:392-424),
      * the class count 1 + max(Y) by a max all-reduce (src/main.cpp:89).
    Host arrays only (runs without a GPU): returns (A_rows, AT_rows, X, Y, info) -- this rank's rows of the backward
    and of the forward matrix as csr_matrix with GLOBAL columns."""
    import os
    from . import datasets as ds
    P, r = dctx.P, dctx.rank
    gpath = os.path.join(dirname, "graph.bin")
    _, _, _, n, m = ds.read_csr_rows(gpath, 0, 0)
    if n != m:
        raise ValueError("graph.bin must be square")
    if n >= 2 ** 31:
        raise ValueError("vertex numbers must fit 31 bits (the transpose exchange packs them as int32)")
    p = partition_bounds(n, P)
    rb, re = p[r], p[r + 1]
    ip, ix, dv, _, _ = ds.read_csr_rows(gpath, rb, re)
    # column normalisation: D[c] = sum over ALL rows
    part = np.bincount(ix.astype(np.int64), weights=dv.astype(np.float64), minlength=n)
    deg = dctx.host_all_reduce(part).astype(np.float32) if P > 1 else part.astype(np.float32)
    dvn = (dv / deg[ix.astype(np.int64)]).astype(np.float32)
    A_rows = csr_matrix(ip, ix, dvn, n)                                     # rows rb..re of A D^-1
    # transpose by exchange: entry (i, c) -> owner of c as (c - p[owner], i, value)
    row_g = (np.repeat(np.arange(re - rb, dtype=np.int64), np.diff(ip.astype(np.int64))) + rb)
    owner = np.searchsorted(np.asarray(p[1:], dtype=np.int64), ix.astype(np.int64), side="right")
    order = np.argsort(owner, kind="stable")                                # CSR (row-ascending) order kept per owner
    cuts = np.searchsorted(owner[order], np.arange(P + 1))
    pieces = []
    for s_ in range(P):
        sel = order[cuts[s_]:cuts[s_ + 1]]
        # 12 bytes per entry (three int32: local column, global row, value bits) -- vertex numbers fit (papers100M: 1.1e8);
        # as int64 triples this exchange moved 2.8 GB for the Reddit shape
        pieces.append(np.stack([(ix[sel].astype(np.int64) - p[s_]).astype(np.int32), row_g[sel].astype(np.int32),
                                dvn[sel].view(np.int32)], axis=1))
    got = dctx.host_all_to_all(pieces) if P > 1 else pieces
    ent = np.concatenate(got, axis=0) if got else np.zeros((0, 3), dtype=np.int32)
    t_order = np.argsort(ent[:, 0], kind="stable")                          # sources arrive in rank = row order
    t_ip = np.zeros(re - rb + 1, dtype=np.int64)
    np.cumsum(np.bincount(ent[:, 0], minlength=re - rb), out=t_ip[1:])
    AT_rows = csr_matrix(t_ip.astype(np.uint32), ent[t_order, 1].astype(np.uint32),
                         np.ascontiguousarray(ent[t_order, 2]).view(np.float32), n)
    X = ds.read_dense_rows(os.path.join(dirname, "features.bin"), "<f4", rb, re)
    Y = ds.read_dense_rows(os.path.join(dirname, "labels.bin"), "<i4", rb, re)
    ymax = int(Y.max()) if Y.size else 0
    num_labels = 1 + (int(dctx.host_all_reduce(np.array([ymax], dtype=np.int64), "max")[0]) if P > 1 else ymax)
    info = {"n": n, "nnz_local": int(ip[-1]), "features": int(X.shape[1]), "num_labels": num_labels, "p": p,
            "host_bytes": int(ip.nbytes + ix.nbytes + 2 * dv.nbytes + ent.nbytes + X.nbytes)}
    return A_rows, AT_rows, X, Y, info


def load_rank_local(dctx: dist_context, dirname: str, chunks: Optional[int] = None):
    """load_rank_local_host + the device side: (Ad, A_Td, Xd, Yd, info) with Ad / A_Td the backward / forward
    dist_row_csr_matrix of src/main.cpp:148-149 and Xd / Yd this rank's shard of the features / labels."""
    A_rows, AT_rows, X, Y, info = load_rank_local_host(dctx, dirname)
    p, n = info["p"], info["n"]
    Ad = dist_row_csr_matrix(dctx, A_rows, p, p, chunks, row_block=True)
    A_Td = dist_row_csr_matrix(dctx, AT_rows, p, p, chunks, row_block=True)
    Xd = dist_row_dn_matrix.__new__(dist_row_dn_matrix)
    Xd.N_, Xd.local = n, dn_matrix.from_numpy(X, dctx.ctx.device)
    Yd = dist_row_dn_matrix.__new__(dist_row_dn_matrix)
    Yd.N_, Yd.local = n, dn_matrix.from_numpy(Y, dctx.ctx.device)
    return Ad, A_Td, Xd, Yd, info


class dist_sparse_linear:
    """reference src/gcn.hpp:50-86 + the pipelined matmul src/cuda_utils.hpp:57-92."""

    def __init__(self, name: str, A: dist_row_csr_matrix, A_T: dist_row_csr_matrix, bcast_buffer,
                 bcast_buffer2, mode: str = "allgather"):
        self.name, self.A, self.A_T = name, A, A_T
        self.bcast = [bcast_buffer, bcast_buffer2]
        self.mode = mode
        self.plans = {}
        self._views = {}
        self._halo_send = None
        self._shared_device = False          # set on first use: more than one rank and overlap on

    def _plan(self, ctx: context, key, M: csr_matrix, d: int):
        pl = self.plans.get(key)
        if pl is None:
            # with more than one rank these SpMMs run while RCCL's kernels share the device: their launch rounds leave (at least)
            # 12 CUs' worth of wave slots free (include/mggcn.h: mggcn_spmm_plan_reserved_cus; +57 % per SpMM without, measured
            # with a stand-in: profiles/experiments/coresident_r04.log)
            shared = self._shared_device
            if shared:
                ctx.lib.mggcn_spmm_plan_reserved_cus(12)
            try:
                pl = self.plans[key] = ops.spmm_plan_for(ctx, M, max(d, 128), d)     # shared across layers
            finally:
                if shared:
                    ctx.lib.mggcn_spmm_plan_reserved_cus(0)
        return pl

    def _run(self, dctx: dist_context, A: dist_row_csr_matrix, tag: str, B: dist_row_dn_matrix,
             C: dist_row_dn_matrix, discard: bool, flags: int) -> None:
        torch = _torch()
        ctx, P, r = dctx.ctx, dctx.P, dctx.rank
        self._shared_device = P > 1 and dctx.overlap
        name = self.name + tag
        beta = 0.0 if discard else 1.0
        d = B.m()
        rows = B.local.n()
        cs = dctx.bcast_stream_id()
        ctx.record(name + "0_matmul-spmm", 0)
        ctx.wait(name + "0_matmul-spmm", cs)                   # comm stream sees the producer of B
        if self.mode == "allgather":
            # the exchange, cut into K pieces of the shard (rows cb[c]..cb[c+1] of every rank): all K
            # all-gathers are queued on the comm stream at once and land one after the other
            cb, K = A.chunk_bounds, len(A.remote_chunks)
            ctx.record(name + "0_matmul-bcast-start", cs)
            # views of the receive buffer / of the shard per piece: built once per (matrix, operand, width) -- at
            # P = 8 the host issues ~30 collectives + ~200 launches per epoch against ~2.4 ms of device work
            vkey = (tag, B.local.buffer(), d)
            views = self._views.get(vkey)
            if views is None:
                views = self._views[vkey] = [(dn_matrix(P * (cb[c + 1] - cb[c]), d, self.bcast[0][P * cb[c] * d:]),
                                              B.local.t[cb[c]:cb[c + 1]]) for c in range(K)]
            pend, gathered = [], []
            for c in range(K):
                g, piece = views[c]
                gathered.append(g)
                # one rank: nothing is remote, nobody reads the gathered copy -- and the "all-gather" would be a 119 MB
                # copy kernel sharing the device with the local SpMM (profiles/r04_forced_dist_summary.md: the sweep
                # round that meets it takes 500-700 us instead of 300)
                pend.append(dctx.all_gather_rows(piece, g.t, cs) if (P > 1 or dctx.self_gather) else None)
            prof = dctx.profile_exchange and (P > 1 or dctx.self_gather)
            if prof:                                           # comm-stream time of the whole exchange
                for c in range(K):
                    pend[c].wait(cs)
                ctx.record(name + "matmul-exchange-end", cs)
                ctx.register_timer(name + "matmul-exchange", name + "0_matmul-bcast-start", name + "matmul-exchange-end")
            # local block first: no dependency on the exchange
            last_local = flags if P == 1 else 0
            ops._spmm(ctx, A.diag, B.local, C.local, self._plan(ctx, (tag, "diag"), A.diag, d), 1.0, beta,
                      last_local)
            for c in range(K):                                 # piece c multiplies while piece c+1 is on the wire
                if prof:                                       # how long the compute stream stalls for piece c
                    ctx.record(name + f"{c}_matmul-bcast-ready", 0)
                    ctx.register_timer(name + f"{c}_matmul-bcast-wait", name + f"{c}_matmul-bcast-ready",
                                       name + f"{c}_matmul-bcast-finish")
                if pend[c] is not None:
                    pend[c].wait(0)
                ctx.record(name + f"{c}_matmul-bcast-finish", 0)
                if P > 1:
                    blk = A.remote_chunks[c]
                    ops._spmm(ctx, blk, gathered[c], C.local, self._plan(ctx, (tag, "remote", c), blk, d), 1.0,
                              1.0, flags if c == K - 1 else 0)
        elif self.mode == "halo":
            h = A.build_halo(ctx.device).halo
            n_send, n_recv = sum(h["send_rows"]), sum(h["recv_rows"])
            if self._halo_send is None or self._halo_send.numel() < max(n_send, 1) * d:
                self._halo_send = torch.empty(max(n_send, 1) * d, dtype=torch.float32, device=ctx.device)
            send = dn_matrix(max(n_send, 1), d, self._halo_send)
            recv = dn_matrix(max(n_recv, 1), d, self.bcast[0])
            ops.gather_rows(ctx, B.local, h["send_idx"], send)            # pack on the compute stream
            ctx.record(name + "0_matmul-halo-packed", 0)
            ctx.wait(name + "0_matmul-halo-packed", cs)
            ctx.record(name + "0_matmul-bcast-start", cs)
            pend = None
            if P > 1:
                pend = dctx.all_to_all_rows(send.t[:n_send], recv.t[:n_recv], h["send_rows"], h["recv_rows"], cs)
                if dctx.profile_exchange:
                    pend.wait(cs)
                    ctx.record(name + "matmul-exchange-end", cs)
                    ctx.register_timer(name + "matmul-exchange", name + "0_matmul-bcast-start", name + "matmul-exchange-end")
            ops._spmm(ctx, A.diag, B.local, C.local, self._plan(ctx, (tag, "diag"), A.diag, d), 1.0, beta,
                      flags if P == 1 else 0)
            if P > 1:
                if dctx.profile_exchange:
                    ctx.record(name + "0_matmul-bcast-ready", 0)
                    ctx.register_timer(name + "0_matmul-bcast-wait", name + "0_matmul-bcast-ready", name + "0_matmul-bcast-finish")
                pend.wait(0)
                ctx.record(name + "0_matmul-bcast-finish", 0)
                blk = h["remote"]
                ops._spmm(ctx, blk, recv, C.local, self._plan(ctx, (tag, "halo"), blk, d), 1.0, 1.0, flags)
        else:  # reference schedule: round i = broadcast shard i || SpMM with block (r, i)
            bufs = [dn_matrix(rows, d, self.bcast[0]), dn_matrix(rows, d, self.bcast[1])]
            for i in range(P):
                if i > 1:
                    ctx.wait(name + f"{i - 1}_matmul-spmm", cs)      # double-buffer hazard (:66-67)
                ctx.record(name + f"{i}_matmul-bcast-start", cs)
                pend = dctx.broadcast_rows(B.local.t, bufs[i % 2].t, i, cs)
                if dctx.profile_exchange:
                    pend.wait(cs)
                    ctx.record(name + f"{i}_matmul-exchange-end", cs)
                    ctx.register_timer(name + f"{i}_matmul-exchange", name + f"{i}_matmul-bcast-start", name + f"{i}_matmul-exchange-end")
                    ctx.record(name + f"{i}_matmul-bcast-ready", 0)
                    ctx.register_timer(name + f"{i}_matmul-bcast-wait", name + f"{i}_matmul-bcast-ready", name + f"{i}_matmul-bcast-finish")
                pend.wait(0)
                ctx.record(name + f"{i}_matmul-bcast-finish", 0)
                blk = A.blocks[i]
                ops._spmm(ctx, blk, bufs[i % 2], C.local, self._plan(ctx, (tag, i), blk, d), 1.0,
                          beta if i == 0 else 1.0, flags if i == P - 1 else 0)
                ctx.record(name + f"{i + 1}_matmul-spmm", 0)
        ctx.record(name + "end_matmul-spmm", 0)
        ctx.register_timer(name + "matmul-spmm", name + "0_matmul-spmm", name + "end_matmul-spmm")

    def __call__(self, dctx, B, C, discard: bool = True, flags: int = 0) -> None:
        self._run(dctx, self.A, "0_", B, C, discard, flags)

    def backward(self, dctx, G, G_out, discard: bool = True) -> None:
        self._run(dctx, self.A_T, "1_", G, G_out, discard, 0)


class dist_row_linear:
    """reference src/gcn.hpp:191-296: replicated W/b, row-sharded X; G_b and G_W are
    summed over ranks (one fused all-reduce instead of the reference's two)."""

    def __init__(self, dctx: dist_context, name: str, in_: int, out: int, backward_out: bool = True,
                 fused: bool = False):
        self.name = name
        self.W, self.b = repl_dn_matrix(dctx, in_, out), repl_dn_matrix(dctx, 1, out)
        # G_W and G_b live in ONE buffer: a single in-place all-reduce, no packing copies
        off_b = (in_ * out + 3) // 4 * 4                       # keep G_b 16-byte aligned
        # ... and four more floats after G_b: the model's LAST layer carries the epoch's two loss sums through its gradient
        # all-reduce there (dist_gcn.train_step: no collective of their own)
        off_t = (off_b + out + 3) // 4 * 4
        self.G_flat = _torch().zeros(off_t + 4, dtype=_torch().float32, device=dctx.ctx.device)
        self.tail = self.G_flat[off_t:off_t + 2]
        self.G_W = repl_dn_matrix.__new__(repl_dn_matrix)
        self.G_W.local = dn_matrix(in_, out, self.G_flat)
        self.G_b = repl_dn_matrix.__new__(repl_dn_matrix)
        self.G_b.local = dn_matrix(1, out, self.G_flat[off_b:])
        _torch().cuda.current_stream().synchronize()            # torch zeroes on ITS stream (the padding takes part in the sum);
                                                                # the kernels run on the context's
        self._grad_pending = None
        self._dctx = dctx
        self.backward_out, self.fused = backward_out, fused
        self.W.init(dctx)
        self.b.init(dctx, _SQRT_1_3)
        self.X = None
        self.ones = None
        self.mW = self.vW = self.mb = self.vb = None
        self.step = 0

    def setX(self, X): self.X = X

    def __call__(self, dctx: dist_context, X: dist_row_dn_matrix, XW: dist_row_dn_matrix,
                 discard: bool = True) -> None:
        ctx, n = dctx.ctx, self.name
        if self.fused and discard:
            ctx.record(n + "0_0_matmul-gemm", 0)
            ops.linear_forward(ctx, X.local, self.W.local, self.b.local, XW.local)
        else:
            ops.broadcast_rows(ctx, self.b.local, XW.local, discard)
            ctx.record(n + "0_0_matmul-gemm", 0)
            ops.matmul(ctx, X.local, self.W.local, XW.local, 1.0, 1.0)
        ctx.record(n + "0_1_matmul-gemm", 0)
        ctx.register_timer(n + "0_matmul-gemm", n + "0_0_matmul-gemm", n + "0_1_matmul-gemm")
        self.X = X

    def backward(self, dctx: dist_context, G: dist_row_dn_matrix, G_out: Optional[dist_row_dn_matrix],
                 discard: bool = True, mask: Optional[dist_row_dn_matrix] = None) -> None:
        ctx, n = dctx.ctx, self.name
        if self.ones is None or self.ones.m() != G.local.n():
            self.ones = dn_matrix(1, G.local.n(), device=ctx.device)
            ctx.fill(self.ones, 1.0)
        ctx.record(n + "1_0_matmul-gemm", 0)
        if self.fused:
            ops.linear_backward_weights(ctx, self.X.local, G.local, self.G_W.local, self.G_b.local)
        else:
            ops.matmul(ctx, self.ones, G.local, self.G_b.local, 1.0, 0.0)
            ops.matmul(ctx, self.X.local, G.local, self.G_W.local, 1.0, 0.0, True)
        # summed over ranks on the comm stream while the backward pass goes on; awaited by
        # finish_backward() (end of dist_gcn.backward) / adam_update
        self._grad_pending = dctx.all_reduce_sum_async(self.G_flat, 0)
        ctx.record(n + "1_2_matmul-gemm", 0)
        if self.backward_out and mask is not None:             # leaky_relu' of the layer below in the epilogue
            ops.matmul_lrelu_backward(ctx, G.local, self.W.local, mask.local, G_out.local, 1.0, False, True)
        elif self.backward_out:
            ops.matmul(ctx, G.local, self.W.local, G_out.local, 1.0, 0.0 if discard else 1.0, False, True)
        ctx.record(n + "1_3_matmul-gemm", 0)
        ctx.register_timer(n + "1_matmul-gemm", n + "1_0_matmul-gemm", n + "1_3_matmul-gemm")

    def finish_backward(self, dctx: dist_context) -> None:
        if self._grad_pending is not None:
            self._grad_pending.wait(0)
            self._grad_pending = None

    def adam_state(self, ctx: context) -> None:
        if self.mW is None:
            d = self._dctx
            self.mW, self.vW = repl_dn_matrix(d, self.W.shape()), repl_dn_matrix(d, self.W.shape())
            self.mb, self.vb = repl_dn_matrix(d, self.b.shape()), repl_dn_matrix(d, self.b.shape())
            for t in (self.mW, self.vW, self.mb, self.vb):
                t.zero(d)
            self.step = 0

    def adam_tensors(self, weight_decay: float):
        return [(self.W.local, self.G_W.local, self.mW.local, self.vW.local, weight_decay),
                (self.b.local, self.G_b.local, self.mb.local, self.vb.local, 0.0)]

    def adam_update(self, dctx: dist_context, lr, beta1, beta2, weight_decay, eps) -> None:
        ctx = dctx.ctx
        self.finish_backward(dctx)
        self.adam_state(ctx)
        self.step += 1
        bc1 = float(np.float32(1 - beta1 ** self.step))
        bc2 = float(np.float32(1 - beta2 ** self.step))
        n = self.name
        ctx.record(n + "0_adam-update", 0)
        W, GW, b, Gb = self.W.local, self.G_W.local, self.b.local, self.G_b.local
        if self.fused:
            ops.adam_fused(ctx, W, GW, self.mW.local, self.vW.local, lr, beta1, beta2, weight_decay, bc1, bc2, eps)
            ops.adam_fused(ctx, b, Gb, self.mb.local, self.vb.local, lr, beta1, beta2, 0.0, bc1, bc2, eps)
        else:
            ops.axpy(ctx, W, GW, weight_decay)
            ops.axpby(ctx, GW, self.mW.local, 1 - beta1, beta1)
            ops.axpby(ctx, Gb, self.mb.local, 1 - beta1, beta1)
            ops.aaxpby(ctx, GW, self.vW.local, 1 - beta2, beta2)
            ops.aaxpby(ctx, Gb, self.vb.local, 1 - beta2, beta2)
            ops.adam_final(ctx, W, self.mW.local, self.vW.local, lr, bc1, bc2, eps)
            ops.adam_final(ctx, b, self.mb.local, self.vb.local, lr, bc1, bc2, eps)
        ctx.record(n + "1_adam-update", 0)
        ctx.register_timer(n + "adam-update", n + "0_adam-update", n + "1_adam-update")

    def get_b(self): return self.b
    def get_W(self): return self.W
    def get_G_W(self): return self.G_W
    def get_G_b(self): return self.G_b


class dist_gcn_layer:
    """reference src/gcn.hpp:520-637 (row_partition = true)."""

    def __init__(self, dctx: dist_context, name: str, A: dist_row_csr_matrix, A_T: dist_row_csr_matrix,
                 in_: int, out: int, activation: bool, residual_layer: bool = False, backward_spmm: bool = True,
                 HW_buffer=None, bcast_buffer=None, bcast_buffer2=None, fused: bool = False,
                 mode: str = "allgather"):
        torch = _torch()
        P, dev = dctx.P, dctx.ctx.device
        self.name = name
        self.A = dist_sparse_linear(name, A, A_T, bcast_buffer, bcast_buffer2, mode)
        self.lin = dist_row_linear(dctx, name, in_, out, backward_spmm, fused)
        self.residual_layer = bool(residual_layer)            # gcn.hpp:527-553
        self.res_lin = dist_row_linear(dctx, name, in_, out, backward_spmm, False) if residual_layer and in_ != out else None
        mn = min(in_, out)
        self.AHW_buffer = torch.empty(max(A.n() * out, A_T.n() * in_) // P, dtype=torch.float32, device=dev)
        self.HW = dist_row_dn_matrix(dctx, A.m(), mn, HW_buffer)
        self.AHW = dist_row_dn_matrix(dctx, A.n(), out, self.AHW_buffer)
        self.G_HW = dist_row_dn_matrix(dctx, A_T.n(), mn, HW_buffer)
        self.G_out = dist_row_dn_matrix(dctx, A_T.n(), in_, self.AHW_buffer)
        self.activation, self.backward_spmm, self.fused = activation, backward_spmm, fused
        self.H = None
        self.mask_input_grad = self.grad_premasked = False      # see gcn.link_fused_backward

    def gemm_first(self) -> bool:
        return self.HW.m() == self.AHW.m()

    def __call__(self, dctx: dist_context, H: dist_row_dn_matrix) -> dist_row_dn_matrix:
        ctx, n = dctx.ctx, self.name
        self.H = H
        act_done = False
        if self.HW.m() == self.AHW.m():
            self.lin(dctx, H, self.HW)
            if self.fused and self.activation:
                self.A(dctx, self.HW, self.AHW, True, MGGCN_SPMM_LEAKY_RELU)
                act_done = True
            else:
                self.A(dctx, self.HW, self.AHW)
        else:
            self.A(dctx, H, self.HW)
            self.lin(dctx, self.HW, self.AHW)
        if self.activation and not act_done:
            ctx.record(n + "0_0_activation", 0)
            ops.leaky_relu_forward(ctx, self.AHW.local, self.AHW.local)
            ctx.record(n + "0_1_activation", 0)
            ctx.register_timer(n + "0_activation", n + "0_0_activation", n + "0_1_activation")
        if self.res_lin is not None:
            self.res_lin(dctx, H, self.AHW, False)
        elif self.residual_layer:
            ops.axpy(ctx, H.local, self.AHW.local, 1.0)
        return self.AHW

    def backward(self, dctx: dist_context, G: dist_row_dn_matrix) -> dist_row_dn_matrix:
        ctx, n = dctx.ctx, self.name
        T = G
        if self.activation and not self.grad_premasked:
            ctx.record(n + "1_0_activation", 0)
            ops.leaky_relu_backward(ctx, self.AHW.local, G.local, self.AHW.local)
            ctx.record(n + "1_1_activation", 0)
            ctx.register_timer(n + "1_activation", n + "1_0_activation", n + "1_1_activation")
            T = self.AHW
        if self.HW.m() == self.AHW.m():
            G_HW = self.G_HW
            if self.backward_spmm:
                self.A.backward(dctx, T, G_HW)
            else:
                G_HW = T
            self.lin.backward(dctx, G_HW, self.G_out, mask=self.H if self.mask_input_grad else None)
            G_out = self.G_out
        else:
            self.lin.setX(self.H)
            self.lin.backward(dctx, T, self.G_HW)
            G_out = self.G_HW
            if self.backward_spmm:
                self.A.backward(dctx, self.G_HW, self.G_out)
                G_out = self.G_out
        if self.res_lin is not None:
            self.res_lin.backward(dctx, G, G_out, False)
        elif self.residual_layer:
            ops.axpy(ctx, G.local, G_out.local, 1.0)
        return G_out

    def linears(self):
        return [self.lin] + ([self.res_lin] if self.res_lin is not None else [])

    def finish_backward(self, dctx) -> None:
        for lin in self.linears():
            lin.finish_backward(dctx)

    def adam_update(self, dctx, lr, beta1, beta2, weight_decay, eps):
        for lin in self.linears():
            lin.adam_update(dctx, lr, beta1, beta2, weight_decay, eps)

    def b(self): return self.lin.get_b()
    def W(self): return self.lin.get_W()
    def GW(self): return self.lin.get_G_W()
    def Gb(self): return self.lin.get_G_b()


class dist_row_softmax_cross_entropy_loss:
    """reference src/gcn.hpp:872-935: everything is row-local; the gradient is scaled
    by the GLOBAL n (:908); loss / accuracy are the sums of the per-rank scalars (:929).
    The reference sums them on the host of its single process; here a 2-float
    all-reduce does it."""

    def __init__(self, name: str, copy: bool = True, fused: bool = False):
        self.inner = softmax_cross_entropy_loss(name, copy, fused, host_sums=False)     # all-reduced as a device tensor

    def __call__(self, dctx: dist_context, H: dist_row_dn_matrix, Y: dist_row_dn_matrix, sync: bool = True):
        self.inner(dctx.ctx, H.local, Y.local, n_global=Y.n(), sync=False)
        self._G = _wrap_local(self.inner.G, H.n())
        self._n = H.n()
        if not sync:
            return None
        dctx.sync()                                   # the reference blocks here too (:928)
        return self.read(dctx)

    def read(self, dctx: dist_context):
        """global (loss, acc) of the last call; the caller has synchronised"""
        dist = _dist()
        H_n = self._n
        if dctx.backend == "nccl":
            sums = self.inner.sums.clone()
            dist.all_reduce(sums, group=dctx.group)
            sums = sums.cpu()
        else:
            sums = self.inner.sums.detach().cpu()
            dist.all_reduce(sums, group=dctx.group)
        s = sums.numpy()
        n = np.float32(H_n)
        return float(np.float32(s[0]) / n), float(np.float32(s[1]) / n)

    def backward(self) -> dist_row_dn_matrix:
        return self._G


def _wrap_local(dn, n_global):
    w = dist_row_dn_matrix.__new__(dist_row_dn_matrix)
    w.N_, w.local = n_global, dn
    return w


class dist_gcn:
    """reference src/gcn.hpp:997-1056 (row_partition = true): per-GPU HW_buffer and two
    receive buffers shared by all layers (:1016-1021); layers get (A_T, A) (:1023)."""

    def __init__(self, dctx: dist_context, A: dist_row_csr_matrix, A_T: dist_row_csr_matrix,
                 sizes: Sequence[int], residual_layer: bool = False, fused: bool = True, mode: str = "allgather"):
        torch = _torch()
        P, dev = dctx.P, dctx.ctx.device
        self.loss_layer = dist_row_softmax_cross_entropy_loss(f"{len(sizes) - 1}_", residual_layer, fused)
        max_d = max(min(sizes[i], sizes[i + 1]) for i in range(len(sizes) - 1))
        nmax = max(A.n(), A.m())
        self.HW_buffer = torch.empty(nmax * max_d // P, dtype=torch.float32, device=dev)
        # all-gather mode keeps the whole gathered B resident (n x max_d: 119 MB on Reddit,
        # nothing next to 288 GB); rounds mode needs the reference's two shard-sized buffers
        big = nmax * max_d if mode in ("allgather", "halo") else nmax * max_d // P
        self.bcast_buffer = torch.empty(big, dtype=torch.float32, device=dev)
        self.bcast_buffer2 = torch.empty(nmax * max_d // P, dtype=torch.float32, device=dev)
        self.layers_: List[dist_gcn_layer] = []
        for i in range(1, len(sizes)):
            self.layers_.append(dist_gcn_layer(dctx, f"{i - 1}_", A_T, A, sizes[i - 1], sizes[i],
                                               i + 1 < len(sizes), residual_layer, i != 1, self.HW_buffer,
                                               self.bcast_buffer, self.bcast_buffer2, fused, mode))
        link_fused_backward(self.layers_, fused)
        self.fused, self._adam = fused, None
        self._loss_host = None                                 # pinned host copy of the epoch's two global loss sums (train_step)
        # this rank's SpMM plans, built side by side before the first epoch (ops.prebuild_plans) instead of one by one
        # inside it: the diagonal block and the pieces of the schedule that runs, both matrices, both widths
        self._plan_wants = []
        for i in range(1, len(sizes)):
            w = min(sizes[i - 1], sizes[i])
            for M in ([A_T] if i == 1 else [A_T, A]):               # layers get (A_T, A); first layer: no backward SpMM
                parts = list(M.blocks) if mode == "rounds" else \
                    [M.diag] + (list(M.remote_chunks) if mode == "allgather" and P > 1 else [])
                self._plan_wants += [(blk, max(w, 128), w) for blk in parts]

    def __call__(self, dctx, H):
        if self._plan_wants:
            # with more than one rank these SpMMs run while RCCL's kernels share the device: launch rounds that leave (at least) 12
            # CUs' worth of wave slots free (include/mggcn.h: mggcn_spmm_plan_reserved_cus; profiles/r04_forced_dist_summary.md)
            shared = dctx.P > 1 and dctx.overlap
            if shared:
                dctx.ctx.lib.mggcn_spmm_plan_reserved_cus(12)
            try:
                ops.prebuild_plans(dctx.ctx, self._plan_wants)
            finally:
                if shared:
                    dctx.ctx.lib.mggcn_spmm_plan_reserved_cus(0)
            self._plan_wants = []
        for layer in self.layers_:
            H = layer(dctx, H)
        return H

    def train_forward(self, dctx: dist_context, H: dist_row_dn_matrix, Y: dist_row_dn_matrix):
        H = self(dctx, H)
        return self.loss_layer(dctx, H, Y)

    def backward(self, dctx: dist_context) -> None:
        G = self.loss_layer.backward()
        for layer in reversed(self.layers_):
            G = layer.backward(dctx, G)
        for layer in self.layers_:                 # gradients are summed over ranks from here on
            layer.finish_backward(dctx)

    def adam_update(self, dctx, lr, beta1, beta2, weight_decay, eps) -> None:
        if not self.fused:
            for layer in self.layers_:
                layer.adam_update(dctx, lr, beta1, beta2, weight_decay, eps)
            return
        for layer in self.layers_:
            layer.finish_backward(dctx)
        self._adam = adam_update_all(dctx.ctx, [lin for l in self.layers_ for lin in l.linears()], self._adam, lr,
                                     beta1, beta2, weight_decay, eps)

    def train_step(self, dctx: dist_context, H: dist_row_dn_matrix, Y: dist_row_dn_matrix, lr, beta1, beta2,
                   weight_decay, eps):
        """forward + loss + backward + Adam with ONE host synchronisation and the loss all-reduce at the
        end of the epoch (see gcn.train_step); the reference's loop body is src/main.cpp:159-166."""
        torch = _torch()
        out = self(dctx, H)
        self.loss_layer(dctx, out, Y, sync=False)
        # The two loss sums ride on the LAST layer's gradient all-reduce (four spare floats behind [G_W | G_b]) instead of a
        # collective and a device-to-host copy of their own after the epoch's synchronisation: at P = 8 that turn-around was
        # ~0.1 ms of idle GPU per 3.5-ms epoch.  (The reference adds its P managed scalars on the host, src/gcn.hpp:929.)
        st = dctx.ctx.cuda_streams[0]
        last = self.layers_[-1].lin
        with torch.cuda.stream(st):
            last.tail.copy_(self.loss_layer.inner.sums)
        self.backward(dctx)                                    # ... -> finish_backward: the compute stream sees the summed buffers
        self.adam_update(dctx, lr, beta1, beta2, weight_decay, eps)
        if self._loss_host is None:
            self._loss_host = torch.empty(2, dtype=torch.float32, pin_memory=True)
        with torch.cuda.stream(st):
            self._loss_host.copy_(last.tail, non_blocking=True)
        dctx.sync()
        s = self._loss_host.numpy()
        n = np.float32(self.loss_layer._n)
        return float(np.float32(s[0]) / n), float(np.float32(s[1]) / n)

    def layers(self): return self.layers_
