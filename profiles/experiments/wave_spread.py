#!/usr/bin/env python3
"""Diagnostic: how far apart do the one-wave tasks of a sweep launch finish?  (MGGCN_SPMM_STAMPS=1: every wave of
spmm_sweep_pair_kernel records its start / end on the 100 MHz constant clock and its XCC id.)  All waves of a launch
start together and have equal work; their spread at the end bounds how far apart they sweep the column panels, i.e.
the resident L2 window.  Usage: python profiles/experiments/wave_spread.py"""
import os, sys
import numpy as np
os.environ["MGGCN_SPMM_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as g
pkg = g.load_package()

(ip, ix, dv), _, _ = pkg.datasets.synth_reddit_like(1.0, seed=1)
n = ip.shape[0] - 1
A = pkg.csr_matrix(ip, ix, dv, n); A.normalize(True)
A_T = A.transpose()
ctx = pkg.context(0)
d = 128
B = pkg.dn_matrix.from_numpy(np.random.default_rng(0).standard_normal((n, d), dtype=np.float32))
C = pkg.dn_matrix(n, d)
for name, M in (("forward", A_T), ("backward", A)):
    buf = pkg.get_matmul_buffer(ctx, M, B, C)
    for _ in range(3):
        pkg.matmul(ctx, M, B, C, buf, 1.0, 0.0)
    ctx.sync()
    S = ctx.lib.mggcn_spmm_plan_num_slices(buf.handle)
    T = buf.num_sweep_tasks()
    print(f"{name}: {S} slices, {T} tasks in all")
    for s_ in range(S):
        out = np.zeros(3 * T, dtype=np.uint64)
        k = ctx.lib.mggcn_spmm_plan_read_stamps(buf.handle, s_, out.ctypes.data, T)
        st = out[:3 * k].reshape(k, 3)
        st = st[st[:, 1] > 0]
        t0, t1 = st[:, 0].astype(np.int64), st[:, 1].astype(np.int64)
        xcc = (st[:, 2] & np.uint64(0xF)).astype(np.int64)
        blk = ((st[:, 2] >> np.uint64(4)) & np.uint64(0xFFFFF)).astype(np.int64)
        hw = (st[:, 2] >> np.uint64(32)).astype(np.int64)
        slot, simd, cu, se = hw & 0xF, (hw >> 4) & 3, (hw >> 8) & 0xF, (hw >> 13) & 7
        # only the LAST launch of the slice is fully recorded per task; tasks of earlier launches keep their stamps too
        # (each task runs once per call): group by launch = task index // round
        base = t0.min()
        dur_all = (t1.max() - base) / 100.0
        print(f"  slice {s_}: {len(st)} tasks, first start -> last end {dur_all:.1f} us")
        rt = 4096          # resident one-wave tasks per launch: 256 CUs x 4 workgroups x 4 waves
        for L in range((len(st) + rt - 1) // rt):
            sel = slice(L * rt, min((L + 1) * rt, len(st)))
            a0, a1 = t0[sel], t1[sel]
            if len(a0) == 0:
                continue
            span = (a1.max() - a0.min()) / 100.0
            ends = (a1 - a0.min()) / 100.0
            starts = (a0 - a0.min()) / 100.0
            q = np.percentile(ends, [0, 5, 50, 95, 100])
            print(f"    launch {L}: {len(a0)} waves, span {span:.1f} us; starts p50 {np.median(starts):.1f} p95 {np.percentile(starts,95):.1f} max {starts.max():.1f}; "
                  f"ends min {q[0]:.1f} p5 {q[1]:.1f} p50 {q[2]:.1f} p95 {q[3]:.1f} max {q[4]:.1f} us "
                  f"(min..max = {100*(q[4]-q[0])/span:.1f} %, p5..p95 = {100*(q[3]-q[1])/span:.1f} % of the launch)")
            if s_ == 0 and L == 1:
                dur = (a1 - a0) / 100.0
                def by(key, name):
                    ks = np.unique(key)
                    print(f"      wave time by {name}: " + "  ".join(f"{k}:{np.median(dur[key == k]):.1f}" for k in ks[:16]))
                by(slot[sel], "HW wave slot"); by(simd[sel], "SIMD"); by(blk[sel] // 256, "blockIdx // 256"); by(np.arange(len(dur)) % 4, "wave in block")
                by(se[sel], "SE"); by(cu[sel], "CU in SE")
                cuid = xcc[sel] * 1000 + se[sel] * 16 + cu[sel]
                med = np.array([np.median(dur[cuid == c]) for c in np.unique(cuid)])
                print(f"      per-CU median wave time: min {med.min():.1f} p50 {np.median(med):.1f} max {med.max():.1f} us over {len(med)} CUs; "
                      f"within-CU spread (p95-p5) median {np.median([np.percentile(dur[cuid == c], 95) - np.percentile(dur[cuid == c], 5) for c in np.unique(cuid)]):.1f} us")
            per = [np.median(ends[xcc[sel] == x]) for x in range(8) if (xcc[sel] == x).any()]
            cnt = [int((xcc[sel] == x).sum()) for x in range(8)]
            mism = int(((blk[sel] % 8) != ((blk[sel] % 8)[0] - xcc[sel][0] + xcc[sel]) % 8).sum())
            print(f"      per-XCC median end {['%.0f' % v for v in per]}  waves per XCC {cnt}  blocks off the round-robin map {mism}")
    del buf
