"""One GEMM shape of the epoch, a few calls -- target for rocprofv3 --pmc passes.
Usage: python profiles/experiments/one_gemm.py <fwd608|bwd608|fwd128> [calls]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as g
pkg = g.load_package()
ctx = pkg.context(0)
n = 232968
which = sys.argv[1]; calls = int(sys.argv[2]) if len(sys.argv) > 2 else 5
sa, sb, at, bt = {"fwd608": ((n, 608), (608, 128), False, False), "bwd608": ((n, 608), (n, 128), True, False),
                  "fwd128": ((n, 128), (128, 128), False, False)}[which]
rng = np.random.default_rng(0)
A = pkg.dn_matrix.from_numpy(rng.standard_normal(sa, dtype=np.float32))
B = pkg.dn_matrix.from_numpy(rng.standard_normal(sb, dtype=np.float32))
M = sa[1] if at else sa[0]; N = sb[0] if bt else sb[1]
C = pkg.dn_matrix(M, N)
for _ in range(calls): pkg.matmul(ctx, A, B, C, 1.0, 0.0, at, bt)
ctx.sync()
print("done", which)
