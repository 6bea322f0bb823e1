"""Device-memory hygiene (GPU box): (1) 300 epochs of one model must not grow the footprint; (2) twenty models created,
trained two epochs and dropped must give their memory back (plans, scratch, buffers); (3) the same through plans with the
sweep form forced.  Prints free device memory (hipMemGetInfo via torch) at each stage."""
import gc
import importlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module("mg-gcn_amd")


def free_mb():
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0] / 2**20


def model(n, seed):
    (ip, ix, dv) = pkg.datasets.synth_powerlaw_csr(n, n * 40, 3000, seed=seed)
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n, 64), dtype=np.float32)
    Y = rng.integers(0, 7, size=(n, 1)).astype(np.int32)
    G = pkg.gcn(pkg.csr_matrix(ip, ix, dv, n), [64, 128, 128, 7], fused=True)
    return G, pkg.dn_matrix.from_numpy(X), pkg.dn_matrix.from_numpy(Y)


ctx = pkg.context(0)
base = free_mb()
print(f"start: {base:.0f} MiB free")
G, Xd, Yd = model(30000, 1)
for _ in range(5):
    G.train_step(ctx, Xd, Yd, 1e-2, 0.9, 0.999, 5e-4, 1e-8)
a = free_mb()
for _ in range(300):
    G.train_step(ctx, Xd, Yd, 1e-2, 0.9, 0.999, 5e-4, 1e-8)
b = free_mb()
print(f"300 epochs: {a:.0f} -> {b:.0f} MiB free (delta {a - b:+.1f})")
assert abs(a - b) < 8, "the footprint of a training model moved"
del G, Xd, Yd
gc.collect(); torch.cuda.empty_cache()
lo = []
for form in ("default", "sweep"):
    if form == "sweep":
        os.environ.update(MGGCN_SPMM_SWEEP_MIN_NNZ="1", MGGCN_SPMM_SWEEP_MIN_RUN_X10="0")
    for k in range(10):
        G, Xd, Yd = model(20000 + 512 * k, 10 + k)
        for _ in range(2):
            G.train_step(ctx, Xd, Yd, 1e-2, 0.9, 0.999, 5e-4, 1e-8)
        del G, Xd, Yd
        gc.collect(); torch.cuda.empty_cache()
        lo.append(free_mb())
    print(f"{form}: free after each of ten create / train / drop cycles: {lo[-10]:.0f} ... {lo[-1]:.0f} MiB")
end = free_mb()
print(f"end: {end:.0f} MiB free (start {base:.0f})")
# (the first use of the runtime keeps ~200 MiB for good: code objects, streams, the GEMM workspace -- a constant, not a leak)
assert abs(lo[0] - lo[-1]) < 8, "memory was not given back"
print("ok")
