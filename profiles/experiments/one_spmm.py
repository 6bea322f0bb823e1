"""One SpMM shape on the Reddit graph, a few calls -- the target for rocprofv3 --pmc passes.
Usage: python profiles/experiments/one_spmm.py <d> [calls]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as g
pkg = g.load_package()
d = int(sys.argv[1]); calls = int(sys.argv[2]) if len(sys.argv) > 2 else 5
(ip, ix, dv), _, _ = pkg.datasets.synth_reddit_like(1.0, seed=1)
n = ip.shape[0] - 1
A = pkg.csr_matrix(ip, ix, dv, n); A.normalize(True); A = A.transpose()
ctx = pkg.context(0)
B = pkg.dn_matrix.from_numpy(np.random.default_rng(0).standard_normal((n, d), dtype=np.float32))
C = pkg.dn_matrix(n, d)
buf = pkg.get_matmul_buffer(ctx, A, B, C)
for _ in range(calls): pkg.matmul(ctx, A, B, C, buf, 1.0, 0.0)
ctx.sync()
print("done", d, calls, buf.num_sweep_tasks())
