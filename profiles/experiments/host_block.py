"""Where does the host block while it enqueues an epoch?  Wall time of each enqueue step with the GPU busy: a step that
takes milliseconds contains a synchronising call (the GPU then idles while the host catches up afterwards).
Usage: python profiles/experiments/host_block.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as g
pkg = g.load_package()
(ip, ix, dv), X, Y = pkg.datasets.synth_reddit_like(1.0, seed=1)
n = ip.shape[0] - 1
sizes = [X.shape[1], 128, 128, 128, 1 + int(Y.max())]
ctx = pkg.context(0)
G = pkg.gcn(pkg.csr_matrix(ip, ix, dv, n), sizes, fused=True)
Xd, Yd = pkg.dn_matrix.from_numpy(X), pkg.dn_matrix.from_numpy(Y)
for _ in range(3): G.train_step(ctx, Xd, Yd, 1e-2, 0.9, 0.999, 5e-4, 1e-8)
for rep in range(2):
    ctx.sync()
    t = [time.perf_counter()]
    H = Xd
    for layer in G.layers():
        H = layer(ctx, H); t.append(time.perf_counter())
    G.loss_layer(ctx, H, Yd, sync=False); t.append(time.perf_counter())
    G.backward(ctx); t.append(time.perf_counter())
    G.adam_update(ctx, 1e-2, 0.9, 0.999, 5e-4, 1e-8); t.append(time.perf_counter())
    ctx.sync(); t.append(time.perf_counter())
    names = [f"layer {i} forward" for i in range(len(G.layers()))] + ["loss", "backward", "adam", "final sync"]
    print("  ".join(f"{nm} {1e3 * (b - a):.3f} ms" for nm, a, b in zip(names, t, t[1:])), flush=True)
