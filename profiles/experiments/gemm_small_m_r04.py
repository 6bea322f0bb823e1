#!/usr/bin/env python3
"""The dense products of one epoch at a rank's share of the rows (P = 8: M = 29 121, P = 4: 58 242, P = 2: 116 484): time per
product, 64-wide tiles when the 128-wide tiling cannot fill the chip (default) against always 128-wide (MGGCN_GEMM_WIDE_TILES=1)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as ge
pkg = ge.load_package()
ctx = pkg.context(0)
rng = np.random.default_rng(0)
def t(fn, reps=20):
    fn(); fn(); ctx.sync(); ctx.record("a", 0)
    for _ in range(reps): fn()
    ctx.record("b", 0); ctx.sync(); ctx.register_timer("t", "a", "b")
    return ctx.measure("t") / reps * 1e3
for M in (29121, 58242, 116484):
    H = pkg.dn_matrix.from_numpy(rng.standard_normal((M, 128), dtype=np.float32))
    W = pkg.dn_matrix.from_numpy(rng.standard_normal((128, 128), dtype=np.float32))
    b = pkg.dn_matrix.from_numpy(rng.standard_normal((1, 128), dtype=np.float32))
    O = pkg.dn_matrix(M, 128)
    X6 = pkg.dn_matrix.from_numpy(rng.standard_normal((M, 608), dtype=np.float32))
    W6 = pkg.dn_matrix.from_numpy(rng.standard_normal((608, 128), dtype=np.float32))
    W41 = pkg.dn_matrix.from_numpy(rng.standard_normal((128, 48), dtype=np.float32))
    O41 = pkg.dn_matrix(M, 48)
    GW = pkg.dn_matrix(128, 128); Gb = pkg.dn_matrix(1, 128)
    row = {
        "H.W+b [Mx128].[128x128]": t(lambda: pkg.ops.linear_forward(ctx, H, W, b, O)),
        "G.W^T (mask) [Mx128].[128x128]^T": t(lambda: pkg.ops.matmul_lrelu_backward(ctx, H, W, H, O, 1.0, False, True)),
        "X^T G + colsum [128xM].[Mx128]": t(lambda: pkg.ops.linear_backward_weights(ctx, H, O, GW, Gb)),
        "X.W0+b [Mx608].[608x128]": t(lambda: pkg.ops.linear_forward(ctx, X6, W6, b, O)),
        "H.W3 [Mx128].[128x48]": t(lambda: pkg.ops.matmul(ctx, H, W41, O41, 1.0, 0.0)),
    }
    print(f"M={M} wide_tiles={os.environ.get('MGGCN_GEMM_WIDE_TILES', '0')}: " + "  ".join(f"{k}: {v:.1f} us" for k, v in row.items()), flush=True)
