#!/usr/bin/env python3
"""A MODEL of the exchange on one GPU: rank 0's whole epoch of the Reddit-shaped 3x128 model at P = 8 / 4 / 2 (rank_epoch_r04.py), with
every all-gather piece replaced by a DELAY on the comm stream of the time the piece's bytes would take at an assumed receive
bandwidth (one workgroup that sleeps for bytes / BW microseconds -- mggcn_debug_occupy_cus -- then the event the compute stream
waits for).  The data is wrong (nothing arrives); the schedule -- which SpMM waits for which piece, what overlaps what -- is the
product's.  It says how much of the exchange the K-piece schedule hides at a given link rate, not what the links deliver."""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as ge
pkg = ge.load_package()
D = pkg.dist
import torch, torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29641")
dist.init_process_group("gloo", rank=0, world_size=1)

cache = "/tmp/reddit_like_cache.npz"                       # one process per configuration (see the .sh): generate once
if os.path.exists(cache):
    z = np.load(cache)
    ip, ix, dv, X, Y = z["ip"], z["ix"], z["dv"], z["X"], z["Y"]
else:
    (ip, ix, dv), X, Y = pkg.datasets.synth_reddit_like(1.0, seed=1)
    np.savez(cache, ip=ip, ix=ix, dv=dv, X=X, Y=Y)
n = ip.shape[0] - 1
A = pkg.csr_matrix(ip, ix, dv, n); A.normalize(True); A_T = A.transpose()
C = 1 + int(Y.max())


class _After:
    def __init__(self, dctx, ev): self.dctx, self.ev = dctx, ev
    def wait(self, stream_id):
        self.dctx.ctx.lib.mggcn_stream_wait_event(self.dctx.ctx.stream(stream_id), self.ev)


class _Done:
    def wait(self, stream_id): pass


class rank0_of(D.dist_context):
    def __init__(self, P, gbps):
        D.dist_context.__init__(self, overlap=True, device_index=0)
        self.P, self.gbps = P, gbps
        self.never = self.ctx.lib.mggcn_malloc(64)
        self.ctx.lib.mggcn_memset_zero(self.never, 64, self.ctx.stream(0))
        self.events = []
        self.exchange_us = 0.0
    def _delay(self, nbytes, stream_id):
        us = nbytes / (self.gbps * 1e3)                       # GB/s = bytes per ns -> us
        self.exchange_us += us
        if self.gbps > 0 and us >= 1.0:
            self.ctx.lib.mggcn_debug_occupy_cus(self.ctx.stream(stream_id), 1, int(us), self.never)
        ev = self.ctx.lib.mggcn_event_create()
        self.ctx.lib.mggcn_event_record(ev, self.ctx.stream(stream_id))
        self.events.append(ev)
        return _After(self, ev)
    def all_gather_rows(self, shard, out, stream_id):
        return self._delay(out.numel() * 4 * (self.P - 1) / self.P, stream_id)     # what this rank receives
    def all_reduce_sum_async(self, flat, after_stream_id=0):
        cs = self.bcast_stream_id()
        self.ctx.record("__m", after_stream_id); self.ctx.wait("__m", cs)
        return self._delay(2 * flat.numel() * 4 * (self.P - 1) / self.P, cs)        # ring all-reduce: 2 (P-1)/P of the buffer
    def all_reduce_sum(self, tensors, stream_id=0): pass
    def drop_events(self):
        for ev in self.events: self.ctx.lib.mggcn_event_destroy(ev)
        self.events = []


for P in [int(x) for x in os.environ.get("RANK_EPOCH_P", "8,4,2").split(",")]:
    p = D.partition_bounds(n, P)
    sizes = [X.shape[1], 128, 128, 128, (C + P - 1) // P * P]
    for K in [int(x) for x in os.environ.get("EXP_CHUNKS", "1,2,4,8").split(",")]:
        row = []
        for gbps in [float(x) for x in os.environ.get("EXP_GBPS", "1e9,350,250,150").split(",")]:   # "infinite", 7 links x 50 / 36 / 21 GB/s
            dctx = rank0_of(P, gbps)
            Ad, A_Td = D.dist_row_csr_matrix(dctx, A, p, p, K), D.dist_row_csr_matrix(dctx, A_T, p, p, K)
            G = D.dist_gcn(dctx, Ad, A_Td, sizes, fused=True, mode="allgather")
            Xd, Yd = D.dist_row_dn_matrix(dctx, X), D.dist_row_dn_matrix(dctx, Y)
            for _ in range(int(os.environ.get("EXP_WARMUP", "3"))):
                G.train_step(dctx, Xd, Yd, 1e-2, 0.9, 0.999, 5e-4, 1e-8)
            torch.cuda.synchronize()
            dctx.drop_events()
            torch.cuda.synchronize()
            reps = int(os.environ.get("EXP_REPS", "8"))
            dctx.exchange_us = 0.0
            t0 = time.perf_counter()
            for _ in range(reps):
                G.train_step(dctx, Xd, Yd, 1e-2, 0.9, 0.999, 5e-4, 1e-8)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / reps * 1e3
            row.append((gbps, ms, dctx.exchange_us / reps / 1e3))
            dctx.drop_events()
            del G, Ad, A_Td, Xd, Yd, dctx
            torch.cuda.empty_cache()
        base = row[0][1]
        print(f"P={P} K={K}: no exchange {base:.3f} ms | " + " | ".join(
            f"{int(g)} GB/s: epoch {ms:.3f} ms (exchange {ex:.2f} ms, exposed {ms - base:.2f})" for g, ms, ex in row[1:]), flush=True)
dist.destroy_process_group()
