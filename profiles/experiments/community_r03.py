#!/usr/bin/env python3
"""Round 3: does the sweep plan depend on the vertex ORDER?  The reference trains on randomly permuted graphs
(`permuted/<name>`, test/data/prep.py:87-94) but also accepts the unpermuted files.  Same degree sequence as the symmetric
Reddit stand-in, three vertex orders:
   random      the stand-in as generated (weights i.i.d. over the vertex ids)
   by-degree   vertices renumbered by decreasing degree (hubs first: popular columns contiguous, heavy rows adjacent)
   community   64 contiguous communities, 90 % of every vertex's edges inside its own (what an unpermuted social graph looks like)
d = 128 and 41, forward matrix (= backward pattern on a symmetric graph); plan decisions alongside."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as g
pkg = g.load_package()
ctx = pkg.context(0)
import scipy.sparse as sp


def timed(M, d):
    n = M.n()
    B = pkg.dn_matrix.from_numpy(np.random.default_rng(0).standard_normal((n, d), dtype=np.float32))
    C = pkg.dn_matrix(n, d)
    buf = pkg.get_matmul_buffer(ctx, M, B, C, max_d=128)
    for _ in range(2): pkg.matmul(ctx, M, B, C, buf, 1.0, 0.0)
    ts = []
    for _ in range(4):
        ctx.sync(); ctx.record("a", 0)
        for _ in range(5): pkg.matmul(ctx, M, B, C, buf, 1.0, 0.0)
        ctx.record("b", 0); ctx.sync(); ctx.register_timer("t", "a", "b")
        ts.append(ctx.measure("t") / 5)
    desc = buf.describe()
    del buf, B, C
    return float(np.median(ts)), desc


def fwd(ip, ix, dv):
    n = len(ip) - 1
    A = pkg.csr_matrix(ip, ix, dv.copy(), n); A.normalize(True)
    return A.transpose()


(ip, ix, dv), _, _ = pkg.datasets.synth_reddit_like(1.0, seed=1, symmetric=True)
n = len(ip) - 1
S = sp.csr_matrix((dv, ix, ip.astype(np.int64)), shape=(n, n))
cases = [("random", (ip, ix, dv))]
deg = np.diff(ip.astype(np.int64))
perm = np.argsort(-deg, kind="stable")
Sd = sp.csr_matrix(S[perm][:, perm]); Sd.sort_indices()
cases.append(("by-degree", (Sd.indptr.astype(np.uint32), Sd.indices.astype(np.uint32), Sd.data)))
# community graph: same n, same nnz budget; vertex v in community v // (n / 64); 90 % of the edges inside
rng = np.random.default_rng(3)
K = 64
size = n // K
m = (int(ip[-1]) - n) // 2
w = rng.pareto(1.3, size=n) + 1.0
w = np.clip(w * ((2 * m) / w.sum()), 1.0, 18000.0)
stubs = np.repeat(np.arange(n, dtype=np.int64), np.maximum(1, np.rint(w)).astype(np.int64))
a = stubs[rng.integers(0, stubs.shape[0], size=int(m * 1.15))]
inside = rng.random(a.shape[0]) < 0.9
# partner: a degree-weighted vertex of the same community (inside) or of the whole graph
order = np.argsort(stubs // size, kind="stable")
sc = stubs[order]
cb = np.searchsorted(sc // size, np.arange(K + 1))
com = np.minimum(a // size, K - 1)
b = np.where(inside, sc[cb[com] + (rng.random(a.shape[0]) * (cb[com + 1] - cb[com])).astype(np.int64)], stubs[rng.integers(0, stubs.shape[0], size=a.shape[0])])
lo, hi = np.minimum(a, b), np.maximum(a, b)
keys = np.unique(lo[lo != hi] * n + hi[lo != hi])[:m]
u, v = keys // n, keys % n
C = sp.coo_matrix((np.ones(len(u), np.float32), (u, v)), shape=(n, n)).tocsr()
C = sp.csr_matrix(C + C.T + sp.eye(n, dtype=np.float32, format="csr")); C.sort_indices(); C.data[:] = 1.0
cases.append(("community", (C.indptr.astype(np.uint32), C.indices.astype(np.uint32), C.data)))
for name, (a_, b_, c_) in cases:
    M = fwd(a_, b_, c_)
    for d in (128, 41):
        ms, desc = timed(M, d)
        print(f"{name:10s} nnz={M.nnz()} d={d:3d}: {ms:.3f} ms   [{desc[60:260]}]", flush=True)
    del M
