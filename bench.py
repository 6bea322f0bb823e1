#!/usr/bin/env python3
"""bench.py -- the reference's headline benchmark on MI355X.

Metric (BASELINE.json): epoch ms of the Reddit 3x128 GCN (sizes 608-128-128-128-41,
4 SpMM-bearing layers, 7 SpMMs per epoch) + the SpMM's achieved HBM GB/s against the
gfx950 roofline, at 1/2/4/8 GPUs.  A "step" is one epoch exactly as the reference
times it (src/main.cpp:122-129 / :159-166): train_forward + backward + adam_update +
device sync.  The dataset cannot be downloaded here, so the graph is the synthetic
Reddit-shaped stand-in of SURVEY.md 8(d) (n = 232 968, nnz = 114 848 860 with
self-loops, heavy-tailed degrees up to 21 657, random columns as after the reference's
vertex permutation), features N(0,1) [n x 608], 41 uniform classes.

  python bench.py [--gpus N --steps K --warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

One process per GPU; N > 1 shards the vertices (1D row partition) and exchanges
feature shards over RCCL.  Total work is fixed as N grows -> "scaling": "strong".
Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

HBM_PEAK_GBPS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_ACHIEVABLE_GBPS = 6300.0    # same guide: measured float4 copy (report against both)


def spmm_bytes_alg(n_rows, n_cols, nnz, d, beta_nonzero=False):
    """indptr + (index,value) stream + B read once + C written once."""
    return 4 * (n_rows + 1) + 8 * nnz + 4 * n_cols * d + 4 * n_rows * d * (2 if beta_nonzero else 1)


def spmm_bytes_gather(n_rows, nnz, d):
    return 4 * (n_rows + 1) + 8 * nnz + 4 * nnz * d + 4 * n_rows * d


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the graph (debug only; 1.0 = Reddit shape)")
    ap.add_argument("--hidden", type=int, nargs="*", default=[128, 128, 128])
    ap.add_argument("--mode", default="allgather", choices=["allgather", "halo", "rounds"])
    ap.add_argument("--no-overlap", action="store_true", help="the reference's -S flag")
    ap.add_argument("--unfused", action="store_true", help="reference launch sequence, no fused kernels")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if args.gpus > 1 and world == 1:
            sys.exit("bench.py --gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    P = world
    pkg = ge.load_package()
    pkg._lib.require_gpu()                     # loud: there is no CPU path
    # Rehearsal on a one-GPU box: MGGCN_BENCH_REHEARSAL=1 maps every rank to cuda:0 and exchanges
    # through gloo (RCCL refuses two ranks on one device).  Exercises this file's N > 1 code path;
    # its numbers mean nothing.
    rehearsal = os.environ.get("MGGCN_BENCH_REHEARSAL", "0") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if P > 1:
        import torch.distributed as dist
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    # ---- workload ------------------------------------------------------------------
    t_gen = time.time()
    (indptr, indices, data), X, Y = pkg.datasets.synth_reddit_like(args.scale, seed=1)
    n, nnz = int(indptr.shape[0] - 1), int(indptr[-1])
    num_labels = 1 + int(Y.max())
    sizes = [X.shape[1]] + list(args.hidden) + [num_labels]
    if P > 1:
        sizes[-1] = (sizes[-1] + P - 1) // P * P            # reference src/main.cpp:135
    fused = not args.unfused
    A = pkg.csr_matrix(indptr, indices, data, n)

    if P == 1:
        ctx = pkg.context(local_rank)
        G = pkg.gcn(A, sizes, fused=fused)                   # normalises + transposes (gcn.hpp:946-948)
        Xd, Yd = pkg.dn_matrix.from_numpy(X), pkg.dn_matrix.from_numpy(Y)
        fwd_mats = {"0_": G.A_T}                             # forward SpMM operand (for byte counts)

        def epoch():          # forward + loss + backward + Adam + sync (src/main.cpp:122-129), one host sync
            return G.train_step(ctx, Xd, Yd, 1e-2, 0.9, 0.999, 5e-4, 1e-8)
        spmm_shape = (G.A_T.n(), G.A_T.m(), G.A_T.nnz())
    else:
        D = pkg.dist
        dctx = D.dist_context(overlap=not args.no_overlap, device_index=local_rank)
        ctx = dctx.ctx
        A.normalize(True)                                    # src/main.cpp:143-144
        A_T = A.transpose()
        p = D.partition_bounds(n, P)
        Ad = D.dist_row_csr_matrix(dctx, A, p, p)
        A_Td = D.dist_row_csr_matrix(dctx, A_T, p, p)
        G = D.dist_gcn(dctx, Ad, A_Td, sizes, fused=fused, mode=args.mode)
        Xd = D.dist_row_dn_matrix(dctx, X)
        Yd = D.dist_row_dn_matrix(dctx, Y)

        def epoch():
            return G.train_step(dctx, Xd, Yd, 1e-2, 0.9, 0.999, 5e-4, 1e-8)
        spmm_shape = (A_Td.diag.n(), n, A_Td.diag.nnz() + A_Td.remote.nnz())
    t_gen = time.time() - t_gen

    # the d = 128 SpMM timers of one epoch (reference timer names, src/gcn.hpp:32-35, :43-46)
    d_main = args.hidden[0]
    nl = len(sizes) - 1
    spmm_timers = []
    for li in range(nl):
        din, dout = sizes[li], sizes[li + 1]
        w = min(din, dout)                       # width the SpMM runs at (gcn.hpp:439-446)
        if w == d_main:
            spmm_timers.append(f"{li}_0_matmul-spmm")
            if li != 0:
                spmm_timers.append(f"{li}_1_matmul-spmm")

    def barrier():
        if P > 1:
            dist.barrier()
        torch.cuda.synchronize()

    losses = []
    for _ in range(args.warmup):
        losses.append(epoch()[0])
    spmm_ms = []
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses.append(epoch()[0])
        spmm_ms.append([ctx.measure(t) for t in spmm_timers])     # hipEventElapsedTime, microseconds of host time
    barrier()
    t1 = time.perf_counter()
    ms = (t1 - t0) * 1000.0 / max(args.steps, 1)
    if P > 1:
        t = torch.tensor([ms], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ms = float(t.item())

    spmm_ms = np.asarray(spmm_ms, dtype=np.float64)
    spmm_avg = float(spmm_ms.mean()) if spmm_ms.size else float("nan")
    nr, nc, nz = spmm_shape
    b_alg = spmm_bytes_alg(nr, nc, nz, d_main)
    b_gather = spmm_bytes_gather(nr, nz, d_main)
    achieved = b_alg / (spmm_avg * 1e-3) / 1e9 if spmm_avg > 0 else float("nan")
    traffic = None
    tf = os.path.join(ROOT, "profiles", "spmm_hbm_traffic.json")     # written from a --pmc pass, if any
    if P == 1 and os.path.exists(tf):
        try:
            traffic = json.load(open(tf)).get("bytes_per_launch")
        except Exception:
            traffic = None

    out = {
        "metric": "epoch_ms (Reddit-shaped 3x128 GCN, full-graph, fp32)",
        "value": round(ms, 4), "unit": "ms", "n_gpus": P, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms, 4), "higher_is_better": False, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "reddit_like_3x128_gcn" if args.scale == 1.0 else f"reddit_like_scale_{args.scale}",
                   "n": n, "nnz": nnz, "sizes": sizes, "spmm_per_epoch": 2 * nl - 1,
                   "parallelism": f"rows{P}" + ("" if P == 1 else f"-{args.mode}"), "fused": fused},
        "roofline": {"bound": "hbm", "kernel": f"spmm_csr_f32 d={d_main}", "achieved": round(achieved, 2),
                     "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 5),
                     "frac_of_achievable_6300": round(achieved / HBM_ACHIEVABLE_GBPS, 5),
                     "traffic": traffic, "ms_per_launch": round(spmm_avg, 4), "bytes_alg": b_alg,
                     "gather_GBps": round(b_gather / (spmm_avg * 1e-3) / 1e9, 1) if spmm_avg > 0 else None,
                     "launches_timed": int(spmm_ms.size)},
        "loss_first_last": [round(float(losses[0]), 5), round(float(losses[-1]), 5)] if losses else None,
        "setup_s": round(t_gen, 1),
    }

    if rank == 0 and P == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(pkg, indptr, indices, data, n, X, Y, sizes, d_main)

    if P > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


def cpu_baseline(pkg, indptr, indices, data, n, X, Y, sizes, d):
    """The oracle (CPU restatement, kind "port") on this box's host cores, bounded to
    roughly 10-30 s: one d-wide forward SpMM on the full graph, then -- if the projected
    time allows -- one full training epoch; otherwise an epoch on a row-scaled graph."""
    orc = ge.load_oracle()
    # the box gives one GPU's share of the host (16 cores); OpenMP would otherwise start one
    # thread per visible hardware thread and oversubscribe the cgroup
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, int(os.environ.get("MGGCN_CPU_BASELINE_THREADS", "16"))))
    orc.lib().orc_set_num_threads(cores)
    A = orc.Csr(indptr, indices, data.copy(), n)
    orc.normalize(A, True)
    A_T = orc.transpose(A)
    B = np.random.default_rng(0).standard_normal((n, d), dtype=np.float32)
    orc.spmm(A_T, B[: A_T.m])                    # warm-up (page faults, thread pool)
    t = time.perf_counter(); orc.spmm(A_T, B); t_spmm = time.perf_counter() - t
    nl = len(sizes) - 1
    projected = t_spmm * (2 * nl - 1) * 1.6      # SpMMs dominate; GEMM/elementwise ~ +60 % on CPU
    res = {"unit": "ms", "cores": cores, "kind": "port",
           "spmm_ms": round(t_spmm * 1e3, 1),
           "spmm_GBps_alg": round(spmm_bytes_alg(n, n, A.nnz, d) / t_spmm / 1e9, 2)}
    if projected <= 45.0:
        O = orc.Gcn(orc.Csr(indptr, indices, data, n), sizes)
        t = time.perf_counter()
        O.train_forward(X, Y); O.backward(); O.adam_update()
        res["value"] = round((time.perf_counter() - t) * 1e3, 1)
        res["sample"] = "1 full epoch of the same workload (oracle.Gcn, fp32, OpenMP)"
    else:
        frac = max(0.02, 20.0 / projected)
        (ip, ix, dv), Xs, Ys = pkg.datasets.synth_reddit_like(frac, seed=1)
        O = orc.Gcn(orc.Csr(ip, ix, dv, ip.shape[0] - 1), sizes)
        t = time.perf_counter()
        O.train_forward(Xs, Ys); O.backward(); O.adam_update()
        res["value"] = round((time.perf_counter() - t) * 1e3 / frac, 1)
        res["sample"] = f"1 epoch on a {frac:.3f}-scale graph (same mean degree), time divided by the scale"
    return res


if __name__ == "__main__":
    main()
