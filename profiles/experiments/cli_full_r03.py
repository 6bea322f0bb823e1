#!/usr/bin/env python3
"""Round 3: the north-star CLI (`mg_gcn`, src/main.cpp:113-170) at BASELINE sizes, both stand-in graphs.

  python profiles/experiments/cli_full_r03.py [asym|sym|both] [P8]

Writes the Reddit-shaped dataset in the reference's on-disk format, runs
  mg_gcn -E 8 train <dir> 3 128 128 128                       (C2)
  MGGCN_OVERSUBSCRIBE=1 mg_gcn -P 8 -R 1 -E 3 train ...        (C3 on one GPU: peer-copy transport)
and prints the per-epoch lines + wall times; then the Python path's plan decisions and per-SpMM times
on the same graph (MGGCN_SPMM_PLAN_LOG=1)."""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

BIN = os.path.join(ROOT, "mg-gcn_amd", "bin", "mg_gcn")


def run_cli(args, cwd, env=None, timeout=900):
    e = dict(os.environ)
    e.update(env or {})
    t = time.time()
    r = subprocess.run([BIN] + args, cwd=cwd, env=e, capture_output=True, text=True, timeout=timeout)
    wall = time.time() - t
    print(f"$ mg_gcn {' '.join(args)}  env={env}  rc={r.returncode} wall={wall:.1f}s", flush=True)
    print(r.stderr[-3000:], flush=True)
    return r, wall


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "both"
    do_p8 = "P8" in sys.argv
    pkg = ge.load_package()
    tmp = tempfile.mkdtemp(prefix="mggcn_cli_")
    for name, sym in (("asym", False), ("sym", True)):
        if which not in (name, "both"):
            continue
        t = time.time()
        (ip, ix, dv), X, Y = pkg.datasets.synth_reddit_like(1.0, seed=1, symmetric=sym)
        print(f"[{name}] generated in {time.time() - t:.1f}s: n={len(ip) - 1} nnz={int(ip[-1])}", flush=True)
        d = os.path.join(tmp, "permuted", f"reddit_{name}")
        t = time.time()
        pkg.datasets.write_dataset(d, ip, ix, dv, X, Y)
        print(f"[{name}] written in {time.time() - t:.1f}s", flush=True)
        r, wall = run_cli(["-E", "8", "train", d, "3", "128", "128", "128"], tmp, {"MGGCN_SPMM_PLAN_LOG": "1"})
        if do_p8:
            for mode in ("allgather", "rounds"):
                run_cli(["-P", "8", "-R", "1", "-E", "3", "train", d, "3", "128", "128", "128"], tmp,
                        {"MGGCN_OVERSUBSCRIBE": "1", "MGGCN_DIST_MODE": mode})
        # the Python path on the same graph: plan decisions + per-SpMM time
        os.environ["MGGCN_SPMM_PLAN_LOG"] = "1"
        ctx = pkg.context(0)
        n = len(ip) - 1
        sizes = [X.shape[1], 128, 128, 128, 1 + int(Y.max())]
        t = time.time()
        G = pkg.gcn(pkg.csr_matrix(ip, ix, dv, n), sizes)
        Xd, Yd = pkg.dn_matrix.from_numpy(X), pkg.dn_matrix.from_numpy(Y)
        ctx.sync()
        print(f"[{name}] python model set-up {time.time() - t:.1f}s", flush=True)
        for _ in range(3):
            G.train_step(ctx, Xd, Yd, 1e-2, 0.9, 0.999, 5e-4, 1e-8)
        import torch
        torch.cuda.synchronize()
        names = []
        for li in range(4):
            names.append(f"{li}_0_matmul-spmm")
            if li:
                names.append(f"{li}_1_matmul-spmm")
        per = {k: [] for k in names}
        t0 = time.perf_counter()
        K = 20
        for _ in range(K):
            G.train_step(ctx, Xd, Yd, 1e-2, 0.9, 0.999, 5e-4, 1e-8)
            for k in names:
                per[k].append(ctx.measure(k))
        torch.cuda.synchronize()
        print(f"[{name}] python epoch {(time.perf_counter() - t0) * 1e3 / K:.3f} ms", flush=True)
        for k in names:
            print(f"   {k}: {np.median(per[k]):.4f} ms")
        del G, Xd, Yd
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
