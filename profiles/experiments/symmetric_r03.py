#!/usr/bin/env python3
"""Round 3: why is the SYMMETRIC Reddit stand-in (pattern A = A^T, what the real dataset is) 29 % slower per
d = 128 SpMM than the SURVEY 8(d) stand-in (3.08 vs 2.39 ms, cli_full_r03.log)?  One variable at a time:

  sym            datasets.synth_reddit_like(symmetric=True): power-law rows AND popular columns, rows sorted by column
  sym-shuffled   the same matrix, entries of every row in random order
  asym-fwd/bwd   the SURVEY stand-in's two matrices (random order inside rows)
  asym-*-sorted  the same two matrices, every row sorted by column

  python profiles/experiments/symmetric_r03.py [d ...]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402


def permute_within_rows(ip, ix, dv, mode, seed=0):
    """mode 'shuffle': random order inside every row; 'sort': ascending columns"""
    n = len(ip) - 1
    rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(ip.astype(np.int64)))
    if mode == "sort":
        key = rows * (1 << 32) + ix.astype(np.int64)
    else:
        key = rows * (1 << 32) + np.random.default_rng(seed).integers(0, 1 << 32, size=ix.shape[0], dtype=np.int64)
    o = np.argsort(key, kind="stable")
    return ip, ix[o], dv[o]


def time_spmm(pkg, ctx, M, d, label, reps=20):
    import torch
    n = M.n()
    rng = np.random.default_rng(d)
    Bd = pkg.dn_matrix.from_numpy(rng.standard_normal((M.m(), d), dtype=np.float32))
    C = pkg.dn_matrix(n, d)
    buf = pkg.get_matmul_buffer(ctx, M, Bd, C, max_d=128)
    for _ in range(3):
        pkg.matmul(ctx, M, Bd, C, buf, 1.0, 0.0)
    ctx.sync()
    ts = []
    ctx.register_timer("spmm", "t0", "t1")
    for _ in range(reps):
        ctx.record("t0", 0); pkg.matmul(ctx, M, Bd, C, buf, 1.0, 0.0); ctx.record("t1", 0)
        ctx.sync()
        ts.append(ctx.measure("spmm"))
    print(f"{label:>22s} d={d:3d}: median {np.median(ts):.4f} ms  min {min(ts):.4f}   [{buf.describe()[:160]}]", flush=True)
    del buf, Bd, C
    torch.cuda.empty_cache()
    return float(np.median(ts))


def main():
    ds_ = [int(x) for x in sys.argv[1:]] or [128, 41]
    pkg = ge.load_package()
    ctx = pkg.context(0)
    ds = pkg.datasets

    def both(ip, ix, dv):
        n = len(ip) - 1
        A = pkg.csr_matrix(ip, ix, dv.copy(), n)
        A.normalize(True)
        return A.transpose(), A            # forward, backward

    (ip, ix, dv), _, _ = ds.synth_reddit_like(1.0, seed=1, symmetric=True)
    cases = [("sym", (ip, ix, dv)), ("sym-shuffled", permute_within_rows(ip, ix, dv, "shuffle"))]
    (ip2, ix2, dv2), _, _ = ds.synth_reddit_like(1.0, seed=1)
    cases += [("asym", (ip2, ix2, dv2)), ("asym-sorted", permute_within_rows(ip2, ix2, dv2, "sort"))]
    for name, (a, b, c) in cases:
        fwd, bwd = both(a, b, c)
        for d in ds_:
            time_spmm(pkg, ctx, fwd, d, name + "-fwd")
            time_spmm(pkg, ctx, bwd, d, name + "-bwd")
        del fwd, bwd


if __name__ == "__main__":
    main()
