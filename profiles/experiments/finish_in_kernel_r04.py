#!/usr/bin/env python3
"""Round 4: the slices of a heavy row combined INSIDE the sweep launch by the slice that arrives last (spmm_sweep.hip:
finish_split_rows) against the separate sweep_combine_kernel launch (MGGCN_SPMM_FINISH_IN_KERNEL=0).

1. Hand-off soak: the same SpMM through both forms, the operand B DIFFERENT from call to call (two operands in turn, so a
   stale line of an earlier call's partial sums cannot pass for a fresh one), every word of C compared -- the two forms
   share one combine function and must agree BIT FOR BIT.  beta = 0 and beta = 1 + leaky-ReLU.
2. Time per call of both forms (median of `reps`).

  python profiles/experiments/finish_in_kernel_r04.py [soak calls per case, default 40] [P: rank 0's share of a P-rank job, default 1]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402


def plan(pkg, ctx, M, Bd, C, in_kernel):
    os.environ["MGGCN_SPMM_FINISH_IN_KERNEL"] = "1" if in_kernel else "0"
    buf = pkg.get_matmul_buffer(ctx, M, Bd, C, max_d=128)
    del os.environ["MGGCN_SPMM_FINISH_IN_KERNEL"]
    return buf


def main():
    import torch
    calls = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    P = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    pkg = ge.load_package()
    ctx = pkg.context(0)
    (ip, ix, dv), _, _ = pkg.datasets.synth_reddit_like(1.0, seed=1)
    n = len(ip) - 1
    A = pkg.csr_matrix(ip, ix, dv.copy(), n)
    A.normalize(True)
    mats = {"fwd": A.transpose(), "bwd": A}
    if P > 1:                                   # rank 0's row block (all columns): the shape of a rank's own launches
        rows = (n + P - 1) // P
        for k, M in list(mats.items()):
            mip, mix, mdv = M.indptr[:rows + 1].copy(), M.indices[:M.indptr[rows]].copy(), M.data[:M.indptr[rows]].copy()
            mats[k] = pkg.csr_matrix(mip, mix, mdv, M.m())
    rng = np.random.default_rng(7)
    bad_total = 0
    for name, M in mats.items():
        for d in (128, 41):
            Bs = [pkg.dn_matrix.from_numpy(rng.standard_normal((M.m(), d), dtype=np.float32)) for _ in range(2)]
            C0 = rng.standard_normal((M.n(), d), dtype=np.float32)
            Ca, Cb = pkg.dn_matrix.from_numpy(C0), pkg.dn_matrix.from_numpy(C0)
            fin = plan(pkg, ctx, M, Bs[0], Ca, True)
            sep = plan(pkg, ctx, M, Bs[0], Cb, False)
            assert fin.handle != sep.handle
            bad = 0
            for beta, flags in ((0.0, 0), (1.0, 1)):
                for it in range(calls):
                    Bd = Bs[it & 1]
                    if beta != 0.0:                 # the same old C on both sides
                        Ca.t.copy_(torch.from_numpy(C0)); Cb.t.copy_(torch.from_numpy(C0))
                    pkg.matmul(ctx, M, Bd, Ca, fin, 1.0 + 0.125 * (it & 3), beta, flags, 0.01)
                    pkg.matmul(ctx, M, Bd, Cb, sep, 1.0 + 0.125 * (it & 3), beta, flags, 0.01)
                    ctx.sync()
                    if not torch.equal(Ca.t, Cb.t):
                        bad += 1
                        diff = (Ca.t != Cb.t).nonzero()
                        print(f"  MISMATCH {name} d={d} beta={beta} call {it}: {diff.shape[0]} words, first {diff[0].tolist()}", flush=True)
            ts = {}
            ctx.register_timer("spmm", "t0", "t1")
            for label, buf, C in (("in-launch", fin, Ca), ("combine kernel", sep, Cb)):
                v = []
                for it in range(25):
                    ctx.record("t0", 0); pkg.matmul(ctx, M, Bs[it & 1], C, buf, 1.0, 0.0); ctx.record("t1", 0)
                    ctx.sync()
                    v.append(ctx.measure("spmm"))
                ts[label] = float(np.median(v[5:]))
            print(f"{name} d={d:3d} P={P}: split rows {fin.num_split_rows()}, launches {fin.num_launches(d)} / {sep.num_launches(d)}; "
                  f"{2 * calls} calls compared, {bad} differ; in-launch {ts['in-launch']:.4f} ms, combine kernel {ts['combine kernel']:.4f} ms",
                  flush=True)
            bad_total += bad
            del fin, sep, Bs, Ca, Cb
            torch.cuda.empty_cache()
    print("RESULT:", "bit-identical" if bad_total == 0 else f"{bad_total} calls differ", flush=True)
    return 0 if bad_total == 0 else 1


if __name__ == "__main__":
    sys.exit(main())
