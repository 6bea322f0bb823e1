#!/usr/bin/env python3
"""Compute-side strong scaling of the WHOLE epoch, measured on one GPU: rank 0's share of the Reddit-shaped 3x128 model at
P = 1 / 2 / 4 / 8 (its row block of both matrices cut into diagonal block + K remote pieces, its shard of every activation, the
full replicated weights) with the exchange switched OFF (the collectives are no-ops: results are wrong, the device work is what a
rank does between exchanges).  epoch(P = 1) / epoch(P) is the speed-up the row partition can reach when the exchange is fully hidden
-- what is left below P is kernel efficiency at 1/P of the rows (one launch round per SpMM piece, smaller GEMMs) and launch gaps."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as ge
pkg = ge.load_package()
D = pkg.dist
import torch, torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29631")
dist.init_process_group("gloo", rank=0, world_size=1)

(ip, ix, dv), X, Y = pkg.datasets.synth_reddit_like(1.0, seed=1)
n = ip.shape[0] - 1
A = pkg.csr_matrix(ip, ix, dv, n); A.normalize(True); A_T = A.transpose()
C = 1 + int(Y.max())


class _Done:
    def wait(self, stream_id): pass


class rank0_of(D.dist_context):
    """rank 0 of a P-rank job whose peers do not exist: every collective returns at once"""
    def __init__(self, P):
        D.dist_context.__init__(self, overlap=True, device_index=0)
        self.P = P
    def all_gather_rows(self, shard, out, stream_id): return _Done()
    def broadcast_rows(self, shard, out, root, stream_id): return _Done()
    def all_to_all_rows(self, send, recv, send_rows, recv_rows, stream_id): return _Done()
    def all_reduce_sum(self, tensors, stream_id=0): pass
    def all_reduce_sum_async(self, flat, after_stream_id=0): return _Done()


base = None
for P in [int(x) for x in os.environ.get("RANK_EPOCH_P", "1,2,4,8").split(",")]:
    dctx = rank0_of(P)
    p = D.partition_bounds(n, P)
    sizes = [X.shape[1], 128, 128, 128, (C + P - 1) // P * P]
    Ad, A_Td = D.dist_row_csr_matrix(dctx, A, p, p), D.dist_row_csr_matrix(dctx, A_T, p, p)
    G = D.dist_gcn(dctx, Ad, A_Td, sizes, fused=True, mode="allgather")
    Xd, Yd = D.dist_row_dn_matrix(dctx, X), D.dist_row_dn_matrix(dctx, Y)
    for _ in range(3):
        G.train_step(dctx, Xd, Yd, 1e-2, 0.9, 0.999, 5e-4, 1e-8)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 10
    for _ in range(K):
        G.train_step(dctx, Xd, Yd, 1e-2, 0.9, 0.999, 5e-4, 1e-8)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / K * 1e3
    names = sorted(k for k in dctx.ctx.timers if k.endswith("matmul-spmm") or k.endswith("matmul-gemm"))
    spmm = sum(dctx.measure(k) for k in names if k.endswith("matmul-spmm"))
    gemm = sum(dctx.measure(k) for k in names if k.endswith("matmul-gemm"))
    base = base or ms
    print(f"P={P}: rank-0 epoch {ms:.3f} ms (x{base / ms:.2f} of P=1; ideal x{P})   SpMM calls {spmm:.3f} ms, GEMM calls {gemm:.3f} ms, rest {ms - spmm - gemm:.3f} ms", flush=True)
    del G, Ad, A_Td, Xd, Yd, dctx
    torch.cuda.empty_cache()
dist.destroy_process_group()
