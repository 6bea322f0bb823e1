#!/usr/bin/env python3
"""Experiment: does a popularity-ordered vertex numbering help the sweep SpMM?  The Reddit-shaped graph as generated
(random columns, like the reference's permuted/ datasets) against the same graph with its vertices renumbered by
descending degree (hot columns of the forward matrix contiguous: the first 8192 columns = one L2 hold 35 % of the
non-zeros).  Times d = 128 and d = 41 on both matrices.  Usage: python profiles/experiments/degree_order.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as g
pkg = g.load_package()
import scipy.sparse as sp

(ip, ix, dv), _, _ = pkg.datasets.synth_reddit_like(1.0, seed=1)
n = ip.shape[0] - 1
ctx = pkg.context(0)

def timeit(M, d, reps=5):
    B = pkg.dn_matrix.from_numpy(np.random.default_rng(0).standard_normal((n, d), dtype=np.float32))
    C = pkg.dn_matrix(n, d)
    buf = pkg.get_matmul_buffer(ctx, M, B, C, max_d=128)
    for _ in range(2): pkg.matmul(ctx, M, B, C, buf, 1.0, 0.0)
    ctx.sync(); ctx.record("a", 0)
    for _ in range(reps): pkg.matmul(ctx, M, B, C, buf, 1.0, 0.0)
    ctx.record("b", 0); ctx.sync(); ctx.register_timer("t", "a", "b")
    ms = ctx.measure("t") / reps
    nl = buf.num_launches(d)
    del buf
    return ms, nl

for order in ("as generated", "degree-sorted"):
    if order == "as generated":
        A = pkg.csr_matrix(ip, ix, dv.copy(), n)
    else:
        t = time.time()
        deg = np.diff(ip.astype(np.int64))
        perm = np.argsort(-deg, kind="stable")                      # new vertex i = old vertex perm[i]
        S = sp.csr_matrix((dv, ix, ip), shape=(n, n))[perm][:, perm]
        S = sp.csr_matrix(S)
        A = pkg.csr_matrix(S.indptr.astype(np.uint32), S.indices.astype(np.uint32), S.data.astype(np.float32), n)
        print(f"permute on host {time.time() - t:.1f} s", flush=True)
    A.normalize(True)
    A_T = A.transpose()
    for env in ([{}] if os.environ.get("EXP_QUICK") else [{}, {"MGGCN_SPMM_PANEL_ROWS": "8192"}, {"MGGCN_SPMM_PANEL_ROWS": "4096"}, {"MGGCN_SPMM_SLICE_MIB": "128"}]):
        for k in ("MGGCN_SPMM_PANEL_ROWS", "MGGCN_SPMM_SLICE_MIB"):
            os.environ.pop(k, None)
        os.environ.update(env)
        f, fl = timeit(A_T, 128); b, bl = timeit(A, 128)
        print(f"{order:14s} {str(env):40s} d=128 fwd {f:.3f} ms ({fl} launches)  bwd {b:.3f} ms ({bl})", flush=True)
    for k in ("MGGCN_SPMM_PANEL_ROWS", "MGGCN_SPMM_SLICE_MIB"):
        os.environ.pop(k, None)
    f, fl = timeit(A_T, 41); b, bl = timeit(A, 41)
    print(f"{order:14s} d=41 fwd {f:.3f} ms ({fl} launches)  bwd {b:.3f} ms ({bl})", flush=True)
