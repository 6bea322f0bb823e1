"""Time of the fused loss pass on the Reddit logits ([232 968 x 41], and 48 / 128 wide), copy = True and False.
Usage: python profiles/experiments/loss_time.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as g
pkg = g.load_package()
ctx = pkg.context(0)
n = 232968
rng = np.random.default_rng(0)
for m in (41, 48, 64, 128):
    H = pkg.dn_matrix.from_numpy(rng.standard_normal((n, m), dtype=np.float32))
    Y = pkg.dn_matrix.from_numpy(rng.integers(0, m, size=(n, 1)).astype(np.int32))
    for copy in (True, False):
        L = pkg.softmax_cross_entropy_loss("t_", copy=copy, fused=True)
        for _ in range(20): L(ctx, H, Y, sync=False)
        ctx.sync(); ctx.record("a", 0)
        for _ in range(50): L(ctx, H, Y, sync=False)
        ctx.record("b", 0); ctx.sync(); ctx.register_timer("t", "a", "b")
        us = ctx.measure("t") / 50 * 1e3
        print(f"m={m:4d} copy={int(copy)}: {us:7.1f} us per loss layer call  ({2 * 4.0 * n * m / us / 1e6:6.2f} TB/s of logits read + gradient written)", flush=True)
