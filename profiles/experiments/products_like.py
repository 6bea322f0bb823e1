#!/usr/bin/env python3
"""Experiment: one SpMM at the ogbn-products shape (n ~ 2.45 M, nnz ~ 126 M,
d = 128; B = 1.25 GB >> Infinity Cache) on ONE GPU: row-split vs sweep, with plan build time."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as ge
pkg = ge.load_package()

n, nnz = 2_449_032, 126_200_000
t = time.time()
ip, ix, dv = pkg.datasets.synth_powerlaw_csr(n, nnz, 17_500, seed=5)
A = pkg.csr_matrix(ip, ix, dv, n); A.normalize(True); A = A.transpose()
print(f"graph {time.time()-t:.1f} s", flush=True)
ctx = pkg.context(0)
d = 128
B = pkg.dn_matrix.from_numpy(np.random.default_rng(0).standard_normal((n, d), dtype=np.float32))
C0, C1 = pkg.dn_matrix(n, d), pkg.dn_matrix(n, d)

def timeit(buf, C, reps=3):
    pkg.matmul(ctx, A, B, C, buf, 1.0, 0.0); ctx.sync()
    ctx.record("a", 0)
    for _ in range(reps): pkg.matmul(ctx, A, B, C, buf, 1.0, 0.0)
    ctx.record("b", 0); ctx.sync(); ctx.register_timer("t", "a", "b")
    return ctx.measure("t") / reps

os.environ["MGGCN_SPMM_ALGO"] = "rowsplit"
t = time.time(); b0 = pkg.get_matmul_buffer(ctx, A, B, C0); tb0 = time.time() - t
print(f"rowsplit: {timeit(b0, C0):.3f} ms (plan {tb0:.1f} s)", flush=True)
os.environ["MGGCN_SPMM_ALGO"] = "sweep"
for mib in (32, 128):
    os.environ["MGGCN_SPMM_SLICE_MIB"] = str(mib)
    t = time.time(); b1 = pkg.get_matmul_buffer(ctx, A, B, C1); tb1 = time.time() - t
    ms = timeit(b1, C1)
    err = float(((C1.t - C0.t).abs().max() / C0.t.abs().max()).item())
    print(f"sweep slice {mib} MiB: {ms:.3f} ms (plan {tb1:.1f} s, {b1.nbytes()/1e6:.0f} MB, tasks {b1.num_sweep_tasks()}) maxrel diff {err:.2e}", flush=True)
    del b1
