#!/usr/bin/env python3
"""Experiment (round 1): how fast does the row-gather SpMM run when every column index
falls inside a window of W rows of B (W * 512 B)?  Same nnz / degree distribution as the
Reddit-shaped graph, columns drawn uniformly from [0, W).  Prices the L2-resident gather
rate that a column-panel sweep could reach.  Prints one line per window."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as ge
pkg = ge.load_package()
import torch

(ip, ix, dv), _, _ = pkg.datasets.synth_reddit_like(1.0, seed=1)
n = ip.shape[0] - 1
ctx = pkg.context(0)
d = int(os.environ.get("EXP_WD", "128"))
B = pkg.dn_matrix.from_numpy(np.random.default_rng(0).standard_normal((n, d), dtype=np.float32))
C = pkg.dn_matrix(n, d)
rng = np.random.default_rng(1)
ALGO = os.environ.get("EXP_ALGO", "rowsplit")
os.environ["MGGCN_SPMM_ALGO"] = ALGO
for W in [int(x) for x in os.environ.get("EXP_WINDOWS", "2048,4096,8192,16384,32768,65536,%d" % n).split(",")]:
    cols = rng.integers(0, W, size=ix.shape[0], dtype=np.uint32)
    A = pkg.csr_matrix(ip, cols, dv, n)
    buf = pkg.get_matmul_buffer(ctx, A, B, C)
    for _ in range(2):
        pkg.matmul(ctx, A, B, C, buf, 1.0, 0.0)
    ctx.sync()
    ctx.record("a", 0)
    for _ in range(5):
        pkg.matmul(ctx, A, B, C, buf, 1.0, 0.0)
    ctx.record("b", 0)
    ctx.sync()
    ctx.register_timer("t", "a", "b")
    ms = ctx.measure("t") / 5
    print(f"{ALGO} window {W:7d} rows = {W*512/2**20:7.2f} MiB : {ms:7.3f} ms/SpMM, gather {ix.shape[0]*4*d/ms/1e9:8.1f} TB/s", flush=True)
    del A, buf
