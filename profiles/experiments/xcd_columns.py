#!/usr/bin/env python3
"""the XCD column partition, MEASURED (MGGCN_SPMM_XCD_COLUMNS=1, plan-side only: every row cut
into 8 column slices with a partial-sum slot each, the task table laid out so that the workgroups dealt to XCD x only
touch slice x of B; spmm_sweep.hip).  d = 128, forward and backward matrix of both Reddit stand-ins; the result is
checked against the default plan's (same sums regrouped: 1e-5 relative).

  python profiles/experiments/xcd_columns.py
"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as g
pkg = g.load_package()
ctx = pkg.context(0)
d = 128


def timed(M, B, C, buf, reps=4, calls=5):
    for _ in range(2): pkg.matmul(ctx, M, B, C, buf, 1.0, 0.0)
    ts = []
    for _ in range(reps):
        ctx.sync(); ctx.record("a", 0)
        for _ in range(calls): pkg.matmul(ctx, M, B, C, buf, 1.0, 0.0)
        ctx.record("b", 0); ctx.sync(); ctx.register_timer("t", "a", "b")
        ts.append(ctx.measure("t") / calls)
    return float(np.median(ts))


for sym in (False, True):
    (ip, ix, dv), _, _ = pkg.datasets.synth_reddit_like(1.0, seed=1, symmetric=sym)
    n = ip.shape[0] - 1
    A = pkg.csr_matrix(ip, ix, dv, n); A.normalize(True)
    A_T = A.transpose()
    B = pkg.dn_matrix.from_numpy(np.random.default_rng(0).standard_normal((n, d), dtype=np.float32))
    for name, M in (("fwd", A_T), ("bwd", A)):
        res = {}
        for label, env in (("default", {}), ("one-slice", {"MGGCN_SPMM_SLICE_MIB": "128"}),
                           ("xcd-columns", {"MGGCN_SPMM_SLICE_MIB": "128", "MGGCN_SPMM_XCD_COLUMNS": "1"})):
            for k in ("MGGCN_SPMM_SLICE_MIB", "MGGCN_SPMM_XCD_COLUMNS"):
                os.environ.pop(k, None)
            os.environ.update(env)
            C = pkg.dn_matrix(n, d)
            buf = pkg.get_matmul_buffer(ctx, M, B, C, max_d=128)        # plans are cached per knob set
            ms = timed(M, B, C, buf)
            res[label] = C.t.clone()
            print(f"{'sym' if sym else 'asym'}-{name} {label:12s} {ms:.3f} ms  launches {buf.num_launches(d)}  {buf.describe()[100:330]}", flush=True)
            del buf, C
        ref = res["default"]
        for label in ("one-slice", "xcd-columns"):
            err = float((res[label] - ref).abs().max() / ref.abs().max())
            print(f"   {label}: max rel diff vs default {err:.2e}", flush=True)
            assert err < 1e-5
