"""Randomised parity sweep of the dense entry points (GPU box): mggcn_gemm_f32 with every transpose combination,
mggcn_gemm_bias_f32, mggcn_gemm_lrelu_bwd_f32 and mggcn_gemm_tn_colsum_f32 over random (M, N, K) -- odd sizes, 1-wide
operands, K long enough for split-K, K shorter than one MFMA step.  Reference: numpy in fp64.  Bar: 1e-4 of the entry's
magnitude budget sum|a||b| (+ |beta||c|), the largest of the output row.

    python3 profiles/experiments/gemm_fuzz_r03.py [cases] [seed]
"""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module("mg-gcn_amd")

SIZES_MN = [1, 2, 3, 4, 5, 7, 8, 15, 16, 17, 31, 32, 33, 40, 41, 47, 48, 63, 64, 65, 100, 127, 128, 129, 130, 200, 255, 256, 257,
            300, 512, 608, 700, 1000, 2049, 5000]
SIZES_K = [1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 31, 32, 33, 41, 64, 100, 127, 128, 129, 300, 608, 1000, 3001, 4096, 4097, 9000,
           30000]


def check(got, want, budget, what, state):
    ok = np.isfinite(got).all()
    scale = np.maximum(budget.max(axis=1, keepdims=True), 1e-30)
    err = float((np.abs(got - want) / scale).max()) if ok else float("inf")
    state["worst"] = max(state["worst"], err)
    if err > 1e-4:
        state["bad"] += 1
        print(f"FAIL {what}: err={err:.3e}", flush=True)


def run(cases=None, seed=None):
    cases = cases if cases is not None else (int(sys.argv[1]) if len(sys.argv) > 1 else 300)
    seed = seed if seed is not None else (int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    rng = np.random.default_rng(seed)
    ctx = pkg.context(0)
    state = {"worst": 0.0, "bad": 0}
    t0 = time.time()
    for case in range(cases):
        M, N = int(rng.choice(SIZES_MN)), int(rng.choice(SIZES_MN))
        K = int(rng.choice(SIZES_K))
        while M * K > 40_000_000 or N * K > 40_000_000:
            K = int(rng.choice(SIZES_K))
        kind = rng.choice(["gemm", "gemm", "bias", "lrelu_bwd", "tn_colsum"])
        f64 = np.float64
        if kind == "gemm":
            A_T, B_T = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
            alpha, beta = [(1.0, 0.0), (1.0, 1.0), (-0.5, 0.25), (2.0, 0.0)][int(rng.integers(0, 4))]
            A = rng.standard_normal((K, M) if A_T else (M, K)).astype(np.float32)
            B = rng.standard_normal((N, K) if B_T else (K, N)).astype(np.float32)
            C0 = rng.standard_normal((M, N)).astype(np.float32)
            if beta == 0.0 and rng.random() < 0.5:
                C0[:] = np.nan
            Ad, Bd, Cd = (pkg.dn_matrix.from_numpy(x) for x in (A, B, C0))
            pkg.matmul(ctx, Ad, Bd, Cd, alpha, beta, A_T, B_T)
            ctx.sync()
            a, b = (A.T if A_T else A).astype(f64), (B.T if B_T else B).astype(f64)
            want, budget = alpha * (a @ b), abs(alpha) * (np.abs(a) @ np.abs(b))
            if beta != 0.0:
                want, budget = want + beta * C0.astype(f64), budget + abs(beta) * np.abs(C0.astype(f64))
            check(Cd.numpy(), want, budget, f"case {case} gemm M={M} N={N} K={K} A_T={A_T} B_T={B_T} alpha={alpha} beta={beta}", state)
        elif kind == "bias":
            X = rng.standard_normal((M, K)).astype(np.float32)
            W = rng.standard_normal((K, N)).astype(np.float32)
            b = rng.standard_normal((1, N)).astype(np.float32)
            out = np.full((M, N), np.nan, np.float32)
            Xd, Wd, bd, od = (pkg.dn_matrix.from_numpy(x) for x in (X, W, b, out))
            pkg.ops.linear_forward(ctx, Xd, Wd, bd, od)
            ctx.sync()
            want = X.astype(f64) @ W.astype(f64) + b.astype(f64)
            budget = np.abs(X.astype(f64)) @ np.abs(W.astype(f64)) + np.abs(b.astype(f64))
            check(od.numpy(), want, budget, f"case {case} bias M={M} N={N} K={K}", state)
        elif kind == "lrelu_bwd":
            A_T, B_T = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
            alpha = float(rng.choice([1.0, -2.0]))
            A = rng.standard_normal((K, M) if A_T else (M, K)).astype(np.float32)
            B = rng.standard_normal((N, K) if B_T else (K, N)).astype(np.float32)
            Z = rng.standard_normal((M, N)).astype(np.float32)
            Z[rng.random((M, N)) < 0.05] = 0.0                        # the boundary: Z > 0 ? 1 : slope
            out = np.full((M, N), np.nan, np.float32)
            Ad, Bd, Zd, od = (pkg.dn_matrix.from_numpy(x) for x in (A, B, Z, out))
            pkg.ops.matmul_lrelu_backward(ctx, Ad, Bd, Zd, od, alpha, A_T, B_T)
            ctx.sync()
            a, bb = (A.T if A_T else A).astype(f64), (B.T if B_T else B).astype(f64)
            mask = np.where(Z > 0, 1.0, 0.01)
            want, budget = alpha * (a @ bb) * mask, abs(alpha) * (np.abs(a) @ np.abs(bb)) * mask
            check(od.numpy(), want, budget, f"case {case} lrelu_bwd M={M} N={N} K={K} A_T={A_T} B_T={B_T}", state)
        else:                                                         # G_W = X^T G, G_b = 1^T G; here M = width of X, K = rows
            X = rng.standard_normal((K, M)).astype(np.float32)
            G = rng.standard_normal((K, N)).astype(np.float32)
            gw = np.full((M, N), np.nan, np.float32)
            gb = np.full((1, N), np.nan, np.float32)
            Xd, Gd, gwd, gbd = (pkg.dn_matrix.from_numpy(x) for x in (X, G, gw, gb))
            pkg.ops.linear_backward_weights(ctx, Xd, Gd, gwd, gbd)
            ctx.sync()
            check(gwd.numpy(), X.astype(f64).T @ G.astype(f64), np.abs(X.astype(f64)).T @ np.abs(G.astype(f64)),
                  f"case {case} tn_colsum G_W M={M} N={N} K={K}", state)
            check(gbd.numpy(), G.astype(f64).sum(axis=0, keepdims=True), np.abs(G.astype(f64)).sum(axis=0, keepdims=True),
                  f"case {case} tn_colsum G_b M={M} N={N} K={K}", state)
        if case % 50 == 49:
            print(f"[{case + 1}/{cases}] worst {state['worst']:.2e} failures {state['bad']} ({time.time() - t0:.0f} s)", flush=True)
    print(f"done: {cases} cases, worst relative error {state['worst']:.3e}, failures {state['bad']}")
    return 1 if state["bad"] else 0


if __name__ == "__main__":
    sys.exit(run())
