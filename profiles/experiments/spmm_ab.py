#!/usr/bin/env python3
"""A/B timing of the sweep SpMM on the Reddit-shaped graph under the current environment (MGGCN_SPMM_* knobs):
d = 128 and d = 41, forward and backward matrix, median of 3 x 5 calls.  Usage: python profiles/experiments/spmm_ab.py [tag]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as g
pkg = g.load_package()
tag = sys.argv[1] if len(sys.argv) > 1 else ""
(ip, ix, dv), _, _ = pkg.datasets.synth_reddit_like(1.0, seed=1, symmetric=os.environ.get('SPMM_AB_SYMMETRIC', '0') == '1')
n = ip.shape[0] - 1
A = pkg.csr_matrix(ip, ix, dv, n); A.normalize(True)
A_T = A.transpose()
ctx = pkg.context(0)
out = []
for d in (128, 41):
    B = pkg.dn_matrix.from_numpy(np.random.default_rng(0).standard_normal((n, d), dtype=np.float32))
    C = pkg.dn_matrix(n, d)
    for name, M in (("fwd", A_T), ("bwd", A)):
        buf = pkg.get_matmul_buffer(ctx, M, B, C, max_d=128)
        for _ in range(2): pkg.matmul(ctx, M, B, C, buf, 1.0, 0.0)
        ts = []
        for _ in range(3):
            ctx.sync(); ctx.record("a", 0)
            for _ in range(5): pkg.matmul(ctx, M, B, C, buf, 1.0, 0.0)
            ctx.record("b", 0); ctx.sync(); ctx.register_timer("t", "a", "b")
            ts.append(ctx.measure("t") / 5)
        out.append(f"d={d} {name} {np.median(ts):.3f} ms ({buf.num_launches(d)} launches)")
        del buf
print(f"{tag:28s} " + "   ".join(out), flush=True)
