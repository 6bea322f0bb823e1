"""Randomised parity sweep of the SpMM entry point (GPU box; not part of the suite -- the suite's cases are fixed).

Every case: random shape (rows, columns), degree law (uniform / power-law / a few giant rows / mostly empty), width d,
alpha / beta / fused leaky-ReLU, and a random plan form (default heuristics, or the sweep form forced with random panel /
slice / permutation knobs -- the knobs are read when a plan is built, so one process can walk through all of them).
Reference: scipy CSR in fp64 on the host.  Bar: 1e-4 relative to the row's magnitude budget sum|a||b| (the tests' bar).

    python3 profiles/experiments/spmm_fuzz_r03.py [cases] [seed]
"""
import importlib
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module("mg-gcn_amd")

WIDTHS = [1, 2, 3, 4, 7, 8, 12, 16, 24, 31, 32, 33, 40, 41, 44, 47, 48, 49, 63, 64, 65, 66, 95, 96, 97, 100, 124, 127, 128,
          129, 130, 132, 192, 200, 255, 256, 257, 300, 512, 608]
KNOBS = ["MGGCN_SPMM_PERMUTE_COLUMNS", "MGGCN_SPMM_SWEEP_MIN_NNZ", "MGGCN_SPMM_PANEL_ROWS", "MGGCN_SPMM_PANEL_ROWS_NARROW",
         "MGGCN_SPMM_SLICE_ROWS", "MGGCN_SPMM_SWEEP_MIN_RUN_X10", "MGGCN_SPMM_ALGO", "MGGCN_SPMM_SWEEP_ROWS_PER_TASK",
         "MGGCN_SPMM_TASKS_PER_WAVE", "MGGCN_SPMM_FAST_PAIRS"]


def random_graph(rng):
    n = int(rng.choice([1, 2, 15, 16, 17, 63, 64, 65, 200, 1000, 1500, 4097, 9000]))
    m = int(rng.choice([1, 2, 31, 64, 65, 300, 1500, 4096, 4097, 20000]))
    law = rng.choice(["uniform", "power", "giants", "sparse", "dense_rows"])
    if law == "uniform":
        lens = rng.integers(0, min(4 * 40, 8 * m) + 1, size=n)
    elif law == "power":
        lens = np.minimum((rng.pareto(1.2, size=n) * 8).astype(np.int64), 6000)
    elif law == "giants":
        lens = rng.integers(0, 12, size=n)
        for r in rng.integers(0, n, size=min(3, n)):
            lens[r] = int(rng.choice([700, 4096, 5000, 20000]))
    elif law == "sparse":
        lens = (rng.random(n) < 0.1) * rng.integers(1, 5, size=n)
    else:
        lens = rng.integers(100, 700, size=n)
    lens = np.asarray(lens, dtype=np.int64)
    ip = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint32)
    nnz = int(ip[-1])
    if rng.random() < 0.5:                                   # duplicates allowed (the reference's files may hold them)
        ix = rng.integers(0, m, size=nnz, dtype=np.uint32)
    else:                                                    # skewed columns: a few hot ones
        ix = np.minimum((rng.pareto(0.8, size=nnz) * max(m / 50, 1)).astype(np.int64), m - 1).astype(np.uint32)
    if rng.random() < 0.5:                                   # sorted rows, as a converter would write them
        for r in range(n):
            ix[ip[r]:ip[r + 1]].sort()
    dv = rng.standard_normal(nnz).astype(np.float32)
    return n, m, ip, ix, dv, law


def random_form(rng):
    for k in KNOBS:
        os.environ.pop(k, None)
    form = rng.choice(["default", "sweep", "sweep", "rowsplit"])
    if form == "sweep":
        os.environ["MGGCN_SPMM_SWEEP_MIN_NNZ"] = "1"
        os.environ["MGGCN_SPMM_SWEEP_MIN_RUN_X10"] = "0"
        os.environ["MGGCN_SPMM_PERMUTE_COLUMNS"] = str(int(rng.integers(0, 2)))
        os.environ["MGGCN_SPMM_PANEL_ROWS"] = str(int(rng.choice([64, 100, 128, 1000, 4096])))
        os.environ["MGGCN_SPMM_PANEL_ROWS_NARROW"] = str(int(rng.choice([64, 96, 160, 4096])))
        os.environ["MGGCN_SPMM_SLICE_ROWS"] = str(int(rng.choice([128, 400, 3000, 100000])))
        if rng.random() < 0.3:
            os.environ["MGGCN_SPMM_SWEEP_ROWS_PER_TASK"] = str(int(rng.choice([1, 4, 8, 16])))
        if rng.random() < 0.2:
            os.environ["MGGCN_SPMM_TASKS_PER_WAVE"] = str(int(rng.choice([2, 3])))
        os.environ["MGGCN_SPMM_FAST_PAIRS"] = str(int(rng.integers(0, 2)))
    elif form == "rowsplit":
        os.environ["MGGCN_SPMM_ALGO"] = "rowsplit"
    return form, {k: os.environ[k] for k in KNOBS if k in os.environ}


def main(cases=None, seed=None):
    saved = {k: os.environ[k] for k in KNOBS if k in os.environ}
    try:
        return run(cases, seed)
    finally:                                                 # leave the environment as it was found (tests import this)
        for k in KNOBS:
            os.environ.pop(k, None)
        os.environ.update(saved)


def run(cases=None, seed=None):
    cases = cases if cases is not None else (int(sys.argv[1]) if len(sys.argv) > 1 else 300)
    seed = seed if seed is not None else (int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    rng = np.random.default_rng(seed)
    ctx = pkg.context(0)
    worst, bad, t0 = 0.0, 0, time.time()
    for case in range(cases):
        n, m, ip, ix, dv, law = random_graph(rng)
        form, env = random_form(rng)
        d = int(rng.choice(WIDTHS))
        alpha, beta = [(1.0, 0.0), (0.5, 2.0), (1.0, 1.0), (-1.5, 0.0), (2.0, -0.5)][int(rng.integers(0, 5))]
        flags = int(rng.integers(0, 2))
        use_plan = rng.random() < 0.85
        d_hint = int(rng.choice(WIDTHS)) if rng.random() < 0.3 else None
        A = pkg.csr_matrix(ip, ix, dv, m)
        B = rng.standard_normal((m, d)).astype(np.float32)
        C0 = rng.standard_normal((n, d)).astype(np.float32)
        if beta == 0.0 and rng.random() < 0.5:
            C0[:] = np.nan                                   # beta == 0 must not read C
        Bd, Cd = pkg.dn_matrix.from_numpy(B), pkg.dn_matrix.from_numpy(C0)
        buf = pkg.get_matmul_buffer(ctx, A, Bd, Cd, alpha, beta, max_d=d_hint) if use_plan else None
        reps = 2 if use_plan else 1                          # the second call re-uses the plan's scratch
        for rep in range(reps):
            if rep:
                Cd = pkg.dn_matrix.from_numpy(C0)
            pkg.matmul(ctx, A, Bd, Cd, buf, alpha, beta, flags)
            ctx.sync()
            got = Cd.numpy()
            S = sp.csr_matrix((dv.astype(np.float64), ix.astype(np.int64), ip.astype(np.int64)), shape=(n, m))
            want = alpha * (S @ B.astype(np.float64))
            # |A| from |values| explicitly: abs(S) would sum duplicate entries first (|sum a| instead of sum |a|)
            S_abs = sp.csr_matrix((np.abs(dv).astype(np.float64), ix.astype(np.int64), ip.astype(np.int64)), shape=(n, m))
            budget = abs(alpha) * (S_abs @ np.abs(B.astype(np.float64)))
            if beta != 0.0:
                want = want + beta * C0.astype(np.float64)
                budget = budget + abs(beta) * np.abs(C0.astype(np.float64))
            if flags:
                want = np.where(want > 0, want, 0.01 * want)
            scale = np.maximum(budget.max(axis=1, keepdims=True) if d else budget, 1e-30)
            ok = np.isfinite(got).all()
            err = float((np.abs(got - want) / scale).max()) if ok and got.size else (0.0 if ok else float("inf"))
            worst = max(worst, err)
            if err > 1e-4:
                bad += 1
                desc = buf.describe() if buf is not None and hasattr(buf, "describe") else "no plan"
                print(f"FAIL case {case} rep {rep}: n={n} m={m} nnz={int(ip[-1])} law={law} d={d} hint={d_hint} alpha={alpha} "
                      f"beta={beta} flags={flags} form={form} env={env} err={err:.3e}\n     {desc}", flush=True)
        if case % 25 == 24:
            print(f"[{case + 1}/{cases}] worst {worst:.2e} failures {bad} ({time.time() - t0:.0f} s)", flush=True)
    print(f"done: {cases} cases, worst relative error {worst:.3e}, failures {bad}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
