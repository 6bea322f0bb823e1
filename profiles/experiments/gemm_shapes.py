"""Times every GEMM shape of one Reddit-shaped epoch (fp32 MFMA kernels, gemm.hip) and prices it
against the fp32 MFMA peak (157 TF) and its own bytes (HBM ~5 TB/s achievable).
Usage: python profiles/experiments/gemm_shapes.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as g
pkg = g.load_package()
ctx = pkg.context(0)
n = 232968
rng = np.random.default_rng(0)
shapes = [  # (name, A shape, B shape, A_T, B_T)
    ("fwd  H.W   608->128", (n, 608), (608, 128), False, False),
    ("fwd  H.W   128->128", (n, 128), (128, 128), False, False),
    ("fwd  H.W   128->41 ", (n, 128), (128, 41), False, False),
    ("bwd  X^T.G 608x128 ", (n, 608), (n, 128), True, False),
    ("bwd  X^T.G 128x128 ", (n, 128), (n, 128), True, False),
    ("bwd  X^T.G 128x41  ", (n, 128), (n, 41), True, False),
    ("bwd  G.W^T 128->128", (n, 128), (128, 128), False, True),
    ("bwd  G.W^T 41->128 ", (n, 41), (128, 41), False, True),
    ("bwd  1^T.G  1x128  ", (n, 1), (n, 128), True, False),
]
tot = 0.0
for name, sa, sb, at, bt in shapes:
    A = pkg.dn_matrix.from_numpy(rng.standard_normal(sa, dtype=np.float32))
    B = pkg.dn_matrix.from_numpy(rng.standard_normal(sb, dtype=np.float32))
    M = sa[1] if at else sa[0]; K = sa[0] if at else sa[1]; N = sb[0] if bt else sb[1]
    C = pkg.dn_matrix(M, N)
    # warm-up long enough for the clock to come back up: building A and B above keeps the GPU idle for ~1 s, and with
    # 3 warm-up calls the 608-wide shapes then timed 15-18 % slow (424-440 us vs 370 us for the same kernel in
    # gemm_timeline.hip after 40 calls; the memory-bound 128-wide shapes did not care)
    for _ in range(60): pkg.matmul(ctx, A, B, C, 1.0, 0.0, at, bt)
    ctx.sync(); ctx.record("a", 0)
    for _ in range(20): pkg.matmul(ctx, A, B, C, 1.0, 0.0, at, bt)
    ctx.record("b", 0); ctx.sync(); ctx.register_timer("t", "a", "b")
    ms = ctx.measure("t") / 20
    fl = 2.0 * M * N * K; by = 4.0 * (M * K + K * N + M * N)
    tot += ms
    print(f"{name}: M={M:6d} N={N:4d} K={K:6d}  {ms*1e3:7.1f} us  {fl/ms/1e9:6.1f} TF  {by/ms/1e6:7.1f} GB/s  "
          f"(floor: {fl/157e12*1e6:5.1f} us mfma, {by/5e12*1e6:5.1f} us hbm)", flush=True)
print(f"sum {tot:.3f} ms")
