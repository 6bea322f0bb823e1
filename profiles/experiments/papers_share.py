#!/usr/bin/env python3
"""Sizing run for BASELINE.json configs[4] (ogbn-papers100M, 8 GPUs) on ONE MI355X: rank 0's share
of the 1D row partition -- a [n/8 x n] CSR block with the public OGB shape (n = 111 059 960 after
padding to x8, ~1.73 G non-zeros with self-loops => 216 M per rank, mean degree 15.6; synthetic
power-law degrees, uniformly random columns: the reference trains on a randomly permuted graph) --
against the FULL all-gathered B [n x 128] fp32 (56.9 GB) resident in HBM, as the all-gather
schedule keeps it.  Reports device memory and the SpMM time of the share."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as ge
pkg = ge.load_package()
import torch

P = 8
n = 111_059_960
rows = n // P
nnz = int(os.environ.get("EXP_NNZ", str(1_730_000_000 // P)))
d = 128
t0 = time.time()
ip, _, dv = pkg.datasets.synth_powerlaw_csr(rows, nnz, 20000, seed=3, self_loops=False)
ix = np.random.default_rng(4).integers(0, n, size=nnz, dtype=np.uint32)
dv = (dv / 15.6).astype(np.float32)
print(f"share: {rows} x {n}, nnz {nnz}, max degree {int(np.diff(ip.astype(np.int64)).max())}  (host gen {time.time()-t0:.0f} s)", flush=True)
A = pkg.csr_matrix(ip, ix, dv, n)
ctx = pkg.context(0)
free0, total = torch.cuda.mem_get_info()
B = pkg.dn_matrix(n, d)
with torch.cuda.stream(ctx.cuda_streams[0]):
    B.t.normal_()
C = pkg.dn_matrix(rows, d)
t0 = time.time()
buf = pkg.get_matmul_buffer(ctx, A, B, C)
print(f"plan: {time.time()-t0:.1f} s host, {buf.nbytes()/1e9:.2f} GB device, sweep tasks {buf.num_sweep_tasks()} (0 = row-split form: "
      f"B is addressed with 64-bit offsets, mean run per panel << 2)", flush=True)
for _ in range(2): pkg.matmul(ctx, A, B, C, buf, 1.0, 0.0)
ctx.sync(); ctx.record("a", 0)
reps = 5
for _ in range(reps): pkg.matmul(ctx, A, B, C, buf, 1.0, 0.0)
ctx.record("b", 0); ctx.sync(); ctx.register_timer("t", "a", "b")
ms = ctx.measure("t") / reps
free1, _ = torch.cuda.mem_get_info()
alg = 4 * (rows + 1) + 8 * nnz + 4 * n * d + 4 * rows * d
print(f"SpMM d={d}: {ms:.2f} ms per rank-share; gathered {nnz*4*d/ms/1e9:.2f} TB/s; algorithmic {alg/1e9:.1f} GB -> {alg/ms/1e6:.0f} GB/s", flush=True)
print(f"device memory in use {(free0-free1)/1e9:.1f} GB of {total/1e9:.0f} GB (B {n*d*4/1e9:.1f} GB, C {rows*d*4/1e9:.1f} GB, CSR {(8*nnz+4*rows)/1e9:.1f} GB + plan)", flush=True)
# spot-check 64 rows against numpy
sel = np.random.default_rng(5).integers(0, rows, size=64)
Bh = None
got = C.t[torch.from_numpy(sel).to(C.t.device)].cpu().numpy()
for k, r in enumerate(sel):
    cols = ix[ip[r]:ip[r + 1]].astype(np.int64)
    vals = dv[ip[r]:ip[r + 1]].astype(np.float64)
    want = (B.t[torch.from_numpy(cols).to(B.t.device)].double().cpu().numpy() * vals[:, None]).sum(0) if len(cols) else np.zeros(d)
    assert np.abs(got[k] - want).max() <= 1e-4 * max(np.abs(want).max(), 1e-6), (r, np.abs(got[k] - want).max())
print("spot check of 64 rows against fp64 numpy: ok", flush=True)
