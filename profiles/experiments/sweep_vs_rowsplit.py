#!/usr/bin/env python3
"""Experiment: row-split SpMM vs column-panel sweep SpMM on the Reddit-shaped graph
(forward matrix, d = 128 and d = 41), several panel widths.  Interleaved rounds in one
process; prints ms per SpMM and the max relative difference between the two results."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as ge
pkg = ge.load_package()
import torch

scale = float(os.environ.get("EXP_SCALE", "1.0"))
(ip, ix, dv), _, _ = pkg.datasets.synth_reddit_like(scale, seed=1)
n = ip.shape[0] - 1
A = pkg.csr_matrix(ip, ix, dv, n)
A.normalize(True)
if os.environ.get("EXP_MATRIX", "AT") == "AT":      # forward matrix (even rows, power-law column popularity)
    A = A.transpose()                              # EXP_MATRIX=A: backward matrix (power-law rows, uniform columns)
ctx = pkg.context(0)
widths = [int(x) for x in os.environ.get("EXP_D", "128,41").split(",")]
panels = [int(x) for x in os.environ.get("EXP_PANELS", "1024,2048,4096").split(",")]


def timeit(buf, B, C, reps=5):
    for _ in range(2):
        pkg.matmul(ctx, A, B, C, buf, 1.0, 0.0)
    ctx.sync()
    ctx.record("a", 0)
    for _ in range(reps):
        pkg.matmul(ctx, A, B, C, buf, 1.0, 0.0)
    ctx.record("b", 0)
    ctx.sync()
    ctx.register_timer("t", "a", "b")
    return ctx.measure("t") / reps


for d in widths:
    B = pkg.dn_matrix.from_numpy(np.random.default_rng(0).standard_normal((n, d), dtype=np.float32))
    C0, C1 = pkg.dn_matrix(n, d), pkg.dn_matrix(n, d)
    os.environ["MGGCN_SPMM_ALGO"] = "rowsplit"
    b0 = pkg.get_matmul_buffer(ctx, A, B, C0)
    t0 = timeit(b0, B, C0)
    print(f"d={d:4d} rowsplit            : {t0:7.3f} ms  (sweep tasks {b0.num_sweep_tasks()})", flush=True)
    os.environ["MGGCN_SPMM_ALGO"] = "sweep"
    for pr in panels:
        os.environ["MGGCN_SPMM_PANEL_ROWS"] = str(pr)
        b1 = pkg.get_matmul_buffer(ctx, A, B, C1)
        t1 = timeit(b1, B, C1)
        err = float(((C1.t - C0.t).abs().max() / C0.t.abs().max()).item())
        print(f"d={d:4d} sweep panel {pr:6d}  : {t1:7.3f} ms  tasks {b1.num_sweep_tasks()} split {b1.num_split_rows()} "
              f"plan {b1.nbytes()/1e6:.0f} MB  maxrel diff vs rowsplit {err:.2e}", flush=True)
        del b1
