"""Split-K sensitivity of the tall reductions G_W = X^T G (M = 608 / 128, N = 128, K = 232 968):
time against the number of K-slices (MGGCN_GEMM_SPLITS) -- are the blocks quantised badly
against the resident slots (256 CUs x 2 or 3 blocks)?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as g
pkg = g.load_package()
ctx = pkg.context(0)
n = 232968
rng = np.random.default_rng(0)
for M in (608, 128):
    A = pkg.dn_matrix.from_numpy(rng.standard_normal((n, M), dtype=np.float32))
    B = pkg.dn_matrix.from_numpy(rng.standard_normal((n, 128), dtype=np.float32))
    C = pkg.dn_matrix(M, 128)
    tiles = (M + 127) // 128
    for splits in [0, 25, 51, 76, 102, 103, 128, 153, 204, 256, 384, 512]:
        if splits: os.environ["MGGCN_GEMM_SPLITS"] = str(splits)
        else: os.environ.pop("MGGCN_GEMM_SPLITS", None)
        for _ in range(3): pkg.matmul(ctx, A, B, C, 1.0, 0.0, True, False)
        ctx.sync(); ctx.record("a", 0)
        for _ in range(20): pkg.matmul(ctx, A, B, C, 1.0, 0.0, True, False)
        ctx.record("b", 0); ctx.sync(); ctx.register_timer("t", "a", "b")
        ms = ctx.measure("t") / 20
        print(f"M={M} splits={'default' if not splits else splits} (blocks {tiles*splits if splits else '?'}): {ms*1e3:.1f} us  {2.0*M*128*n/ms/1e9:.1f} TF", flush=True)
