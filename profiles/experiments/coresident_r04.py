#!/usr/bin/env python3
"""VERDICT r03 item 4, second half: what does a kernel that SHARES the device cost the sweep SpMM, and does leaving it room help?
One GPU.  Rank 0's share of the Reddit-shaped forward matrix at P = 2 / 4 / 8 (diagonal block + K remote pieces, d = 128), timed
  * alone,
  * next to a stand-in for a collective kernel: I workgroups of 256 threads that hold their wave slots for the whole
    measurement (mggcn_debug_occupy_cus; RCCL's channels are such workgroups, the single-rank run of r04_forced_dist showed a
    copy kernel doing it),
  * with launch rounds that leave R compute units' worth of slots free (MGGCN_SPMM_RESERVED_CUS=R at plan time)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as ge
pkg = ge.load_package()
D = pkg.dist

(ip, ix, dv), _, _ = pkg.datasets.synth_reddit_like(1.0, seed=1)
n = ip.shape[0] - 1
A = pkg.csr_matrix(ip, ix, dv, n); A.normalize(True); A = A.transpose()
ctx = pkg.context(0)
lib = ctx.lib
d = 128
Ball = pkg.dn_matrix.from_numpy(np.random.default_rng(0).standard_normal((n, d), dtype=np.float32))
flag = lib.mggcn_malloc(64)                                   # device memory: polling host memory disturbs the launch path itself
zero, one = (ctypes.c_uint32 * 1)(0), (ctypes.c_uint32 * 1)(1)

side = {"high": lib.mggcn_stream_create(1), "low": lib.mggcn_stream_create(0)}
SIDE = os.environ.get("EXP_SIDE_STREAM", "low")

def timed(run, occupy, reps=8):
    run(); run(); ctx.sync()
    lib.mggcn_memcpy_h2d(flag, ctypes.addressof(zero), 4, side["high"]); lib.mggcn_stream_synchronize(side["high"])
    if occupy:
        lib.mggcn_debug_occupy_cus(side[SIDE], occupy, 60000, flag)          # at most 60 ms, on a stream of its own
    ctx.record("a", 0)
    for _ in range(reps): run()
    ctx.record("b", 0)
    lib.mggcn_event_synchronize(ctx.events["b"])
    lib.mggcn_memcpy_h2d(flag, ctypes.addressof(one), 4, side["high"])        # the stand-in leaves
    ctx.sync(); ctx.register_timer("t", "a", "b")
    return ctx.measure("t") / reps

print('side stream priority:', SIDE, flush=True)
for P in (2, 4, 8):
    rows = n // P
    K = 2 if P == 2 else 4
    diag, remote = D.split_local_remote(A, 0, rows)
    cb = D.chunk_bounds(rows, K)
    chunks = D.split_remote_chunks(remote, P, rows, K)
    B0 = pkg.dn_matrix(rows, d, Ball.t)
    Bs = [pkg.dn_matrix(P * (cb[c + 1] - cb[c]), d, Ball.t.view(-1)[P * cb[c] * d:]) for c in range(K)]
    C = pkg.dn_matrix(rows, d)
    for R in (0, 16, 32, 64):
        os.environ["MGGCN_SPMM_RESERVED_CUS"] = str(R)
        pd = pkg.get_matmul_buffer(ctx, diag, B0, C)
        plans = [pkg.get_matmul_buffer(ctx, chunks[c], Bs[c], C) for c in range(K)]
        def run():
            pkg.matmul(ctx, diag, B0, C, pd, 1.0, 0.0)
            for c in range(K):
                pkg.matmul(ctx, chunks[c], Bs[c], C, plans[c], 1.0, 1.0)
        row = [f"{timed(run, I):.3f}" for I in (0, 16, 32, 64, 128)]
        print(f"P={P} K={K} reserved_cus={R:3d} tasks diag/piece {pd.num_sweep_tasks()}/{plans[0].num_sweep_tasks()} launches {pd.num_launches(d)}+{K}x{plans[0].num_launches(d)}"
              f"  ms with 0/16/32/64/128 co-resident workgroups: {' '.join(row)}", flush=True)
os.environ.pop("MGGCN_SPMM_RESERVED_CUS", None)
