"""How long the HOST needs to enqueue one epoch (Python layer -> ctypes -> hipLaunch), against the
GPU time of the same epoch: at P = 8 the per-rank GPU work is ~1/6 of the single-GPU epoch while the
number of calls is the same (more, with the exchange), so the enqueue time must stay well below it.
Backward + Adam contain no host synchronisation: their wall time with an idle queue IS the issue time."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as g
pkg = g.load_package()
scale = float(os.environ.get("EXP_SCALE", "1.0"))
(ip, ix, dv), X, Y = pkg.datasets.synth_reddit_like(scale, seed=1)
n = ip.shape[0] - 1
sizes = [X.shape[1], 128, 128, 128, 1 + int(Y.max())]
ctx = pkg.context(0)
G = pkg.gcn(pkg.csr_matrix(ip, ix, dv, n), sizes, fused=True)
Xd, Yd = pkg.dn_matrix.from_numpy(X), pkg.dn_matrix.from_numpy(Y)
for _ in range(3):
    G.train_forward(ctx, Xd, Yd); G.backward(ctx); G.adam_update(ctx, 1e-2, 0.9, 0.999, 5e-4, 1e-8); ctx.sync()
fw, bw, tot = [], [], []
for _ in range(10):
    t0 = time.perf_counter()
    G.train_forward(ctx, Xd, Yd)               # ends with the loss sync
    t1 = time.perf_counter()
    G.backward(ctx); G.adam_update(ctx, 1e-2, 0.9, 0.999, 5e-4, 1e-8)
    t2 = time.perf_counter()                   # queue was empty at t1: t2 - t1 = pure enqueue time
    ctx.sync()
    t3 = time.perf_counter()
    fw.append(t1 - t0); bw.append(t2 - t1); tot.append(t3 - t0)
med = lambda v: sorted(v)[len(v) // 2] * 1e3
print(f"scale {scale}: epoch {med(tot):.2f} ms; forward incl. loss sync {med(fw):.2f} ms; backward+adam ENQUEUE {med(bw):.3f} ms "
      f"(GPU time of that part {med(tot) - med(fw):.2f} ms)")
