#!/usr/bin/env python3
"""Experiment: compute-side strong scaling of the distributed SpMM, measured on ONE GPU: for
P in {1,2,4,8} build rank 0's (diagonal, merged-remote) pair of the Reddit-shaped forward
matrix and time  C = A_diag B_0 ; C += A_remote B_all  at d = 128.  No communication: this is
the part of the per-rank epoch that the 1D row partition divides by P."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as ge
pkg = ge.load_package()
D = pkg.dist

(ip, ix, dv), _, _ = pkg.datasets.synth_reddit_like(1.0, seed=1)
n = ip.shape[0] - 1
A = pkg.csr_matrix(ip, ix, dv, n); A.normalize(True); A = A.transpose()
ctx = pkg.context(0)
d = 128
Ball = pkg.dn_matrix.from_numpy(np.random.default_rng(0).standard_normal((n, d), dtype=np.float32))

def timeit(fn, reps=5):
    fn(); fn(); ctx.sync()
    ctx.record("a", 0)
    for _ in range(reps): fn()
    ctx.record("b", 0); ctx.sync(); ctx.register_timer("t", "a", "b")
    return ctx.measure("t") / reps

base = None
for P in (1, 2, 4, 8):
    rows = n // P
    diag, remote = D.split_local_remote(A, 0, rows)
    B0 = pkg.dn_matrix(rows, d, Ball.t)                 # rank 0's shard = first rows of the gathered buffer
    C = pkg.dn_matrix(rows, d)
    pd = pkg.get_matmul_buffer(ctx, diag, B0, C)
    pr = pkg.get_matmul_buffer(ctx, remote, Ball, C) if P > 1 else None
    def run():
        pkg.matmul(ctx, diag, B0, C, pd, 1.0, 0.0)
        if P > 1:
            pkg.matmul(ctx, remote, Ball, C, pr, 1.0, 1.0)
    ms = timeit(run)
    base = base or ms
    print(f"P={P}: rank-0 SpMM {ms:.3f} ms  (ideal {base/P:.3f}, efficiency {base/P/ms*100:.0f} %)  "
          f"diag nnz {diag.nnz()} tasks {pd.num_sweep_tasks()}  remote nnz {remote.nnz() if P>1 else 0} tasks {pr.num_sweep_tasks() if pr else 0}", flush=True)
    # the exchange cut into K pieces (dist.split_remote_chunks): K smaller remote SpMMs instead of one
    for K in ((2, 4) if P >= 4 else ()):
        cb = D.chunk_bounds(rows, K)
        chunks = D.split_remote_chunks(remote, P, rows, K)
        Bs = [pkg.dn_matrix(P * (cb[c + 1] - cb[c]), d, Ball.t.view(-1)[P * cb[c] * d:]) for c in range(K)]
        plans = [pkg.get_matmul_buffer(ctx, chunks[c], Bs[c], C) for c in range(K)]
        def run_k():
            pkg.matmul(ctx, diag, B0, C, pd, 1.0, 0.0)
            for c in range(K):
                pkg.matmul(ctx, chunks[c], Bs[c], C, plans[c], 1.0, 1.0)
        msk = timeit(run_k)
        print(f"      K={K} pieces: {msk:.3f} ms ({msk/ms*100-100:+.0f} % vs one remote SpMM)  tasks/piece {plans[0].num_sweep_tasks()}", flush=True)
