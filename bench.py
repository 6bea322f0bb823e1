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
Rank 0 prints ONE JSON line.  Plain `python bench.py --gpus N` (no launcher, WORLD_SIZE
unset) starts the N ranks itself -- torch.distributed.run as a CHILD process, before
this process has touched the GPU -- and relays rank 0's line and the children's exit code:
one command, like the reference (`mg_gcn -P N ...`, README.md:44).  At N > 1 the line also
carries the drop-in CLI's own epoch time on the same files (`cli_epoch_ms` and friends: the
single-process C++ form, one enqueue thread per GPU, RCCL and peer-copy transports).
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
# 256 CUs x 64 B/clk x the clock the chip holds under the SpMM: 2.345 GHz inside the bench's own epochs (GRBM_GUI_ACTIVE / 8
# / kernel duration with no other counter armed, profiles/experiments/epoch_clocks_r03.log; the TCP-counter passes of r02
# slowed the kernel to 2.13 GHz, which is where the earlier 34.9 TB/s came from)
L1_FILL_PEAK_GBPS = 38400.0


def spmm_bytes_alg(n_rows, n_cols, nnz, d, beta_nonzero=False):
    """indptr + (index,value) stream + B read once + C written once."""
    return 4 * (n_rows + 1) + 8 * nnz + 4 * n_cols * d + 4 * n_rows * d * (2 if beta_nonzero else 1)


def spmm_bytes_gather(n_rows, nnz, d):
    return 4 * (n_rows + 1) + 8 * nnz + 4 * nnz * d + 4 * n_rows * d


def spmm_kernel_sha():
    """content hash of the SpMM kernel sources: profiles/spmm_hbm_traffic.json carries the hash its counters were
    collected with, so a stale `traffic` figure is visible in the line (and fails tests/test_abi_host.py)"""
    import hashlib
    h = hashlib.sha256()
    for f in ("spmm.hip", "spmm_sweep.hip", "spmm_internal.h", "plan_host.cpp", "plan_host.h"):
        with open(os.path.join(ROOT, "mg-gcn_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def cli_leg(dataset_dir, cwd, hidden, flags=(), env=None, epochs=8, timeout=300):
    """One run of `mg_gcn [flags] -E <epochs> train <dir> <k> <h...>` (the reference's interface, src/main.cpp:113-131 /
    :134-170): median of the CLI's OWN per-epoch seconds (epochs 2..), its set-up (process start to the end of epoch 0
    minus one epoch: file load, normalise / transpose, partition, plans), the first epoch's loss and -- from the
    MGGCN_TIMING lines -- the transport and enqueue mode the distributed classes ran with.  The child runs in its own
    session and is killed as a group when it exceeds `timeout` (a first multi-GPU contact must not cost the line)."""
    import signal
    import subprocess
    exe = os.path.join(ROOT, "mg-gcn_amd", "bin", "mg_gcn")
    if not os.path.exists(exe):
        return None
    e = dict(os.environ)
    if e.pop("MGGCN_HOST_THREADS_AUTO", None):        # the per-rank share of the cores was meant for N processes; this is one
        e.pop("MGGCN_HOST_THREADS", None)
    e.update(env or {})
    e["MGGCN_TIMING"] = "1"
    cmd = [exe] + list(flags) + ["-E", str(epochs), "train", dataset_dir, str(len(hidden))] + [str(h) for h in hidden]
    t = time.perf_counter()
    proc = subprocess.Popen(cmd, cwd=cwd, env=e, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, start_new_session=True)
    try:
        _, err = proc.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(proc.pid, signal.SIGKILL)
        except OSError:
            pass
        proc.wait()
        return {"error": f"timeout after {timeout} s"}
    wall = time.perf_counter() - t
    if proc.returncode != 0:
        return {"error": err[-300:]}
    ep, info, issue = [], {}, []
    for ln in err.splitlines():
        t4 = ln.split()
        if len(t4) == 4 and t4[0].isdigit():
            ep.append((float(t4[1]), float(t4[3])))
        elif ln.startswith("[mggcn timing] transport ") and len(t4) >= 6:
            info = {"transport": t4[3], "enqueue_threads": int(t4[5])}
        elif ln.startswith("[mggcn timing] epoch ") and len(t4) == 6 and t4[4] == "host-issue-ms":   # wall time minus the wait for the devices
            issue.append(float(t4[5]))
    if len(ep) < 3:
        return {"error": "no epoch lines"}
    med = float(np.median([x[1] for x in ep[2:]]))
    out = {"epoch_ms": round(med * 1e3, 4), "setup_s": round(wall - sum(x[1] for x in ep[1:]) - med, 2),
           "loss_first": ep[0][0], "epochs": len(ep)}
    out.update(info)
    if len(issue) >= 3:
        out["host_issue_ms"] = round(float(np.median(issue[2:])), 4)
    # the CLI's own per-epoch timer dump (csvs/<name>_<sizes>_<P>.csv, reference src/main.cpp:100-111, :168; lines
    # "<epoch>_<rank>_<timer>:<ms>"): rank 0's SpMM and GEMM timers of the LAST epoch -- on the compute stream, so an SpMM's
    # figure includes whatever it waited for its pieces
    try:
        import glob
        last = str(len(ep) - 1)
        spmm = gemm = 0.0
        seen = False
        for path in glob.glob(os.path.join(cwd, "csvs", "*.csv")):
            for ln in open(path):
                name, _, val = ln.strip().rpartition(":")
                parts = name.split("_")
                if len(parts) < 3 or parts[0] != last or parts[1] != "0":
                    continue
                if name.endswith("_matmul-spmm") and parts[-2] in ("0", "1") and len(parts) == 5:     # "<e>_<rank>_<layer>_<0|1>_matmul-spmm"
                    spmm += float(val); seen = True
                elif name.endswith("_matmul-gemm") and len(parts) == 5:
                    gemm += float(val)
            os.remove(path)                              # the next leg writes its own
        if seen:
            out["rank0_spmm_ms_per_epoch"] = round(spmm, 4)
            out["rank0_gemm_ms_per_epoch"] = round(gemm, 4)
    except Exception:                                    # noqa: BLE001 -- a diagnostic, never worth the line
        pass
    return out


def run_cli_epochs(pkg, indptr, indices, data, X, Y, hidden, epochs=8):
    """N = 1: the same workload written in the reference's on-disk format and run through `mg_gcn train ...`."""
    import shutil
    import tempfile
    tmp = tempfile.mkdtemp(prefix="mggcn_bench_cli_")
    try:
        d = os.path.join(tmp, "permuted", "bench")
        pkg.datasets.write_dataset(d, indptr, indices, data, X, Y)
        r = cli_leg(d, tmp, hidden, epochs=epochs, timeout=600)
        if r is None or "error" in r:
            return r
        return {"cli_epoch_ms": r["epoch_ms"], "cli_setup_s": r["setup_s"], "cli_loss_first": r["loss_first"], "cli_epochs": r["epochs"],
                # the reference's canonical run is the default TWENTY epochs (src/main.cpp:52): what that costs end to end here --
                # start-up (files, normalise / transpose, four SpMM plans) dwarfs the training at this size
                "cli_default_run_s": round(r["setup_s"] + 20 * r["epoch_ms"] * 1e-3, 2)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def run_cli_multi_gpu(dataset_dir, P, hidden, mode, overlap, rehearsal, epochs=8):
    """N > 1: the PRODUCT PATH of north_star -- `mg_gcn -P N -R 1 train <dir> ...`, one process driving the N GPUs
    (src/main.cpp:134-170) -- on the files the Python ranks just trained on, after they have let go of the GPUs.  Legs:
      default  one enqueue thread per GPU, RCCL (grouped calls become per-thread calls)        -> cli_epoch_ms, cli_transport, ...
      p2p      same, exchange by peer copies on the copy engines (no compute unit taken from the SpMM)  -> cli_p2p_*
      push     same, the SENDER copies (MGGCN_P2P_PUSH=1: writes over xGMI instead of reads)           -> cli_p2p_push_*
      serial   the reference's process model: ONE host thread issues every GPU's work, RCCL    -> cli_serial_*
      (N > 2)  peer copies pulled on one stream per rank instead of one per xGMI link              -> cli_p2p_one_stream_*
               peer copies, two pieces per SpMM instead of four (half the host calls)              -> cli_p2p_two_pieces_*
    Each leg is bounded (own session, killed on timeout); a failed leg reports its error and the next one still runs."""
    import tempfile
    cwd = tempfile.mkdtemp(prefix="mggcn_bench_cli_")
    base = {"MGGCN_DIST_MODE": mode}
    if rehearsal:
        base["MGGCN_OVERSUBSCRIBE"] = "1"                      # N ranks wrapped over the one GPU: peer-copy transport only
    flags = ["-P", str(P), "-R", "1"] + ([] if overlap else ["-S", "x"])
    threads = {"MGGCN_ENQUEUE_THREADS": "1"}
    legs = [("cli", threads)] if rehearsal else [("cli", threads), ("cli_p2p", dict(threads, MGGCN_COMM_TRANSPORT="p2p")),
                                                 ("cli_serial", {"MGGCN_ENQUEUE_THREADS": "0"})]
    if not rehearsal:                    # ... the copies turned round: senders write into the receivers' buffers (posted writes over xGMI)
        legs.append(("cli_p2p_push", dict(threads, MGGCN_COMM_TRANSPORT="p2p", MGGCN_P2P_PUSH="1")))
    if not rehearsal and P > 2:          # ... and the peer copies pulled on ONE stream per rank, copy after copy (are the per-peer streams worth it?)
        legs.append(("cli_p2p_one_stream", dict(threads, MGGCN_COMM_TRANSPORT="p2p", MGGCN_P2P_PEER_STREAMS="0")))
        # ... and in two pieces per SpMM instead of four: the per-peer form issues ~1500 HIP calls per rank and epoch at P = 8 (counted on
        # the stream model of tests/native) -- is it the host that bounds it?
        legs.append(("cli_p2p_two_pieces", dict(threads, MGGCN_COMM_TRANSPORT="p2p", MGGCN_DIST_CHUNKS="2")))
    out = {}
    try:
        for key, extra in legs:
            r = cli_leg(dataset_dir, cwd, hidden, flags, dict(base, **extra), epochs=epochs, timeout=150)
            if r is None:
                return None
            for k, v in r.items():
                out[f"{key}_{k}"] = v
    finally:
        import shutil
        shutil.rmtree(cwd, ignore_errors=True)
    return out


def multi_gpu_env(env, world):
    """Environment every multi-process / multi-GPU leg needs (set here, not left to the caller):
      HSA_ENABLE_IPC_MODE_LEGACY=0  the host driver of this pool only supports dmabuf IPC; without it RCCL's (and
                                    torch's) cross-process buffer sharing fails with `hipIpcGetMemHandle: invalid argument`
      NCCL_DEBUG=WARN               (unless set) RCCL's warnings go to stderr
      MGGCN_HOST_THREADS            the N ranks share one host: every rank builds up to four SpMM plans side by side and
                                    each plan builder starts MGGCN_HOST_THREADS threads (default: every core it sees)"""
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("NCCL_DEBUG", "WARN")               # RCCL's own warnings on stderr: a first multi-GPU run should say why it failed
    if "MGGCN_HOST_THREADS" not in env:
        try:
            avail = len(os.sched_getaffinity(0))
        except AttributeError:
            avail = os.cpu_count() or 8
        env["MGGCN_HOST_THREADS"] = str(max(2, min(16, avail // (4 * max(world, 1)) or 2)))
        env["MGGCN_HOST_THREADS_AUTO"] = "1"          # chosen here, not by the caller: the CLI legs (one process) drop it again
    return env


def self_launch(n, args):
    """`python bench.py --gpus N` without a launcher: run the N ranks as a child `python -m torch.distributed.run`
    (this process has made no GPU call and makes none), pass rank 0's JSON line through, exit with the child's code.
    If the ranks end WITHOUT a line (a failed or hung first contact of torch.distributed / RCCL with the node: the
    child is ended after MGGCN_BENCH_RANKS_TIMEOUT_S, default 1500 s), the product CLI still gets its run: see
    cli_only_line -- the one-process C++ form over peer copies depends on neither."""
    import socket
    import subprocess
    import threading
    with socket.socket() as sk:                       # a free rendezvous port on the loopback interface
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = multi_gpu_env(dict(os.environ), n)
    import tempfile
    parked = os.path.join(tempfile.gettempdir(), f"mggcn_bench_headline_{os.getpid()}_{port}.json")
    env["MGGCN_BENCH_HEADLINE_FILE"] = parked
    sys.stdout.flush()
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)     # stderr is inherited
    got_line = [False]

    def relay():                                      # the line reaches our stdout as is, the moment it is written
        for line in proc.stdout:
            if line.lstrip().startswith("{") and '"metric"' in line:
                got_line[0] = True
            sys.stdout.write(line)
            sys.stdout.flush()
    th = threading.Thread(target=relay, daemon=True)
    th.start()
    limit = float(os.environ.get("MGGCN_BENCH_RANKS_TIMEOUT_S", "1500"))
    try:
        rc = proc.wait(timeout=limit)
        reason = f"the ranks exited with code {rc} and no line"
    except subprocess.TimeoutExpired:
        proc.terminate()                              # the launcher ends its workers on SIGTERM
        try:
            proc.wait(timeout=60)
        except subprocess.TimeoutExpired:
            proc.kill()
            proc.wait()
        rc, reason = 124, f"the ranks printed no line within {limit:.0f} s"
    th.join(10)
    line = None
    try:
        if os.path.exists(parked):
            with open(parked) as fh:
                line = json.load(fh)
            os.remove(parked)
    except (OSError, ValueError):
        line = None
    if got_line[0] or rc == 0:
        return rc
    if line is not None:                              # the timed epochs had finished: the ranks went down in the extras after them
        sys.stderr.write(f"[bench] {reason}: printing the headline the ranks had measured before\n")
        line["extras_error"] = reason + " (after the timed epochs: comm report / other piece count / CLI legs missing)"
        print(json.dumps(line), flush=True)
        return 0
    sys.stderr.write(f"[bench] {reason}: timing the drop-in CLI alone\n")
    line = cli_only_line(n, args, reason)
    if line is None:
        return rc
    print(json.dumps(line), flush=True)
    return 0


def cli_only_line(n, args, reason):
    """The line of a run whose Python ranks failed: the same synthetic dataset, written in the reference's format by THIS
    process (numpy only, no GPU call), trained by `mg_gcn -P N -R 1 ...` -- `value` is the CLI's own median epoch (the
    peer-copy leg when it ran, else the RCCL leg) and `value_source` says so; the ranks' failure is kept in `ranks_error`."""
    import shutil
    import tempfile
    if args.workload != "reddit_like":
        return None
    rehearsal = os.environ.get("MGGCN_BENCH_REHEARSAL", "0") == "1"
    import torch
    if torch.cuda.device_count() == 0:                # (counting devices initialises nothing) no GPU: nothing to time
        return None
    pkg = ge.load_package()
    tmp = tempfile.mkdtemp(prefix="mggcn_bench_cli_only_")
    try:
        (indptr, indices, data), X, Y = pkg.datasets.synth_reddit_like(args.scale, seed=1, symmetric=args.symmetric)
        n_rows, nnz, feats, labels = int(indptr.shape[0] - 1), int(indptr[-1]), int(X.shape[1]), 1 + int(Y.max())
        pkg.datasets.write_dataset(tmp, indptr, indices, data, X, Y)
        del indptr, indices, data, X, Y
        cli = run_cli_multi_gpu(tmp, n, args.hidden, args.mode, not args.no_overlap, rehearsal, epochs=max(args.steps + 2, 5))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    if not cli:
        return None
    leg = next((k for k in ("cli_p2p", "cli_p2p_push", "cli", "cli_serial") if f"{k}_epoch_ms" in cli), None)
    if leg is None:
        return None
    ms = cli[f"{leg}_epoch_ms"]
    out = {"metric": "epoch_ms (Reddit-shaped 3x128 GCN, full-graph, fp32)", "value": ms, "unit": "ms", "n_gpus": n,
           "steps": cli[f"{leg}_epochs"] - 2, "warmup": 2, "ms_per_step": ms, "higher_is_better": False, "scaling": "strong",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": ("reddit_like" if args.scale == 1.0 else f"reddit_like_scale_{args.scale}") + ("_symmetric" if args.symmetric else ""),
                      "n": n_rows, "nnz": nnz, "sizes": [feats] + list(args.hidden) + [(labels + n - 1) // n * n],
                      "parallelism": f"rows{n}-{args.mode}", "fused": True},
           "value_source": f"mg_gcn -P {n} -R 1 (the drop-in CLI, one process; {leg} leg: {cli.get(leg + '_transport', '?')} transport), "
                           "median of its own per-epoch times -- the Python ranks produced no line",
           "ranks_error": reason, "roofline": None, "cpu_baseline": None}
    out.update(cli)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the graph (debug only; 1.0 = Reddit shape)")
    ap.add_argument("--hidden", type=int, nargs="*", default=[128, 128, 128])
    ap.add_argument("--mode", default="allgather", choices=["allgather", "halo", "rounds"])
    ap.add_argument("--no-overlap", action="store_true", help="the reference's -S flag")
    ap.add_argument("--chunks", type=int, default=0, help="pieces of the all-gather exchange (0 = default: 2 at N = 2, 4 above)")
    ap.add_argument("--unfused", action="store_true", help="reference launch sequence, no fused kernels")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the extra keys (symmetric stand-in epoch, the mg_gcn CLI's own epoch time); never the headline")
    ap.add_argument("--workload", default="reddit_like", choices=["reddit_like", "products_like"],
                    help="reddit_like = BASELINE.json's headline config (default); products_like = configs[3]'s graph shape "
                         "(n = 2 449 032, 126.2 M non-zeros, F = 128, 47 classes) -- manual mode, B >> Infinity Cache: "
                         "the SpMM is HBM-bound there")
    ap.add_argument("--symmetric", action="store_true",
                    help="HEADLINE workload = the symmetric stand-in (pattern A = A^T like the real Reddit); manual mode")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus, args))
    if args.chunks > 0:
        os.environ["MGGCN_DIST_CHUNKS"] = str(args.chunks)
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        multi_gpu_env(os.environ, int(os.environ["WORLD_SIZE"]))
    # The contract is ONE JSON line on stdout.  Libraries write there too (RCCL prints a five-line version banner to
    # stdout when its first communicator comes up; gloo prints its rank table): everything this process writes to
    # file descriptor 1 from here on goes to stderr, and the JSON line goes to the real stdout at the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if os.environ.get("MGGCN_BENCH_FAIL_RANKS") == "1" and world > 1:     # test hook (tests/test_gpu_bench.py): a failed first contact ("2": after the timed epochs)
        sys.exit(3)
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world                          # under a launcher the launcher's world size is the truth
    P = world
    pkg = ge.load_package()
    pkg._lib.require_gpu()                     # loud: there is no CPU path
    # Rehearsal on a one-GPU box: MGGCN_BENCH_REHEARSAL=1 maps every rank to cuda:0 and exchanges
    # through gloo (RCCL refuses two ranks on one device).  Exercises this file's N > 1 code path;
    # its numbers mean nothing.
    rehearsal = os.environ.get("MGGCN_BENCH_REHEARSAL", "0") == "1"
    # MGGCN_BENCH_FORCE_DIST=1: take the N > 1 code path (process group, rank-local load, dist_gcn, comm report) with ONE
    # rank -- the only way to run that path over the real RCCL backend on a one-GPU box (tests/test_gpu_bench.py)
    multi = P > 1 or os.environ.get("MGGCN_BENCH_FORCE_DIST", "0") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if multi:
        import torch.distributed as dist
        if P == 1 and "RANK" not in os.environ:                 # forced single-rank run without a launcher
            os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
                              MASTER_PORT=os.environ.get("MASTER_PORT", "29571"))
        high_priority_comm = False
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            # RCCL's kernels on a HIGH-priority stream, like the reference's communication stream (stream_create(i, 0),
            # src/matrix.hpp:53-60, :82): ProcessGroupNCCL runs its collectives on an internal stream of its own -- the
            # context's comm stream only carries the event edges -- and that stream is of default priority unless asked
            opts = None
            try:
                opts = dist.ProcessGroupNCCL.Options()
                opts.is_high_priority_stream = True
            except Exception:                                   # noqa: BLE001 -- an older torch: default priority
                opts = None
            kw = {"pg_options": opts} if opts is not None else {}
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), **kw)
            high_priority_comm = opts is not None

    # ---- workload ------------------------------------------------------------------
    t_gen = time.time()
    fused = not args.unfused

    def make_workload(symmetric):
        if args.workload == "products_like":
            return pkg.datasets.synth_products_like(args.scale, seed=5, symmetric=True)      # OGB's graph is undirected
        return pkg.datasets.synth_reddit_like(args.scale, seed=1, symmetric=symmetric)
    wl_name = {"reddit_like": "reddit_like_3x128_gcn", "products_like": "products_like_3x128_gcn"}[args.workload]
    if not multi:
        (indptr, indices, data), X, Y = make_workload(args.symmetric)
        n, nnz = int(indptr.shape[0] - 1), int(indptr[-1])
        num_labels = 1 + int(Y.max())
        sizes = [X.shape[1]] + list(args.hidden) + [num_labels]
        A = pkg.csr_matrix(indptr, indices, data, n)

    if not multi:
        ctx = pkg.context(local_rank)
        G = pkg.gcn(A, sizes, fused=fused)                   # normalises + transposes (gcn.hpp:946-948)
        Xd, Yd = pkg.dn_matrix.from_numpy(X), pkg.dn_matrix.from_numpy(Y)
        fwd_mats = {"0_": G.A_T}                             # forward SpMM operand (for byte counts)

        def epoch():          # forward + loss + backward + Adam + sync (src/main.cpp:122-129), one host sync
            return G.train_step(ctx, Xd, Yd, 1e-2, 0.9, 0.999, 5e-4, 1e-8)
        spmm_shape = (G.A_T.n(), G.A_T.m(), G.A_T.nnz())
    else:
        # The synthetic graph is written ONCE in the reference's on-disk format (rank 0; the real Reddit would
        # simply be there) and every rank reads only its own rows of it (dist.load_rank_local): no rank ever holds
        # the whole A, A^T or the other ranks' blocks -- the reference's one process loads everything once
        # (src/main.cpp:82-85); P processes doing the same would hold P copies on one host.
        import shutil
        import tempfile
        D = pkg.dist
        dctx = D.dist_context(overlap=not args.no_overlap, device_index=local_rank)
        ctx = dctx.ctx
        tmp = os.path.join(tempfile.gettempdir(), f"mggcn_bench_{args.workload}_{os.environ.get('MASTER_PORT', '0')}_{args.scale}")
        if rank == 0:
            (indptr, indices, data), X, Y = make_workload(args.symmetric)
            pkg.datasets.write_dataset(tmp, indptr, indices, data, X, Y)
            del indptr, indices, data, X, Y
        dist.barrier()
        Ad, A_Td, Xd, Yd, info = D.load_rank_local(dctx, tmp)
        dist.barrier()                                   # (the files stay: the CLI legs read them after the timed epochs)
        n = info["n"]
        nnz = int(dctx.host_all_reduce(np.array([info["nnz_local"]], dtype=np.int64))[0])
        sizes = [info["features"]] + list(args.hidden) + [(info["num_labels"] + P - 1) // P * P]   # src/main.cpp:135
        G = D.dist_gcn(dctx, Ad, A_Td, sizes, fused=fused, mode=args.mode)

        def epoch():
            return G.train_step(dctx, Xd, Yd, 1e-2, 0.9, 0.999, 5e-4, 1e-8)
        spmm_shape = (A_Td.diag.n(), n, A_Td.diag.nnz() + A_Td.remote.nnz())
    t_gen = time.time() - t_gen

    # the SpMM timers of one epoch by width (reference timer names, src/gcn.hpp:32-35, :43-46): the d = 128
    # calls are the dominant kernel (`roofline`), the logits-width calls the second (`roofline_narrow`)
    d_main = args.hidden[0]
    nl = len(sizes) - 1
    timers_by_width = {}
    for li in range(nl):
        w = min(sizes[li], sizes[li + 1])        # width the SpMM runs at (gcn.hpp:439-446)
        names = timers_by_width.setdefault(w, [])
        names.append(f"{li}_0_matmul-spmm")
        if li != 0:
            names.append(f"{li}_1_matmul-spmm")
    spmm_timers = timers_by_width.get(d_main, [])
    d_narrow = min(sizes[-2], sizes[-1])
    narrow_timers = timers_by_width.get(d_narrow, []) if d_narrow != d_main else []

    def barrier():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    losses = []
    for _ in range(args.warmup):
        losses.append(epoch()[0])
    spmm_ms, narrow_ms = [], []
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses.append(epoch()[0])
        spmm_ms.append([ctx.measure(t) for t in spmm_timers])     # hipEventElapsedTime, microseconds of host time
        narrow_ms.append([ctx.measure(t) for t in narrow_timers])
    barrier()
    t1 = time.perf_counter()
    ms = (t1 - t0) * 1000.0 / max(args.steps, 1)
    if multi:
        t = torch.tensor([ms], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ms = float(t.item())

    def spmm_roofline(ms_lists, d, launches_per_call, traffic_key):
        """achieved = ALGORITHMIC bytes of one SpMM call / its HIP-event time on the compute stream"""
        a = np.asarray(ms_lists, dtype=np.float64)
        if not a.size:
            return None
        avg, med, mn = float(a.mean()), float(np.median(a)), float(a.min())
        nr, nc, nz = spmm_shape
        b_alg = spmm_bytes_alg(nr, nc, nz, d)
        ach = b_alg / (avg * 1e-3) / 1e9
        traffic, src, sha_ok = None, None, None
        tf = os.path.join(ROOT, "profiles", "spmm_hbm_traffic.json")     # written from separate --pmc passes
        if not multi and os.path.exists(tf) and args.workload == "reddit_like" and args.scale == 1.0:   # counters of THAT shape
            try:
                j = json.load(open(tf))
                traffic = j.get(traffic_key, j.get("bytes_per_launch") if traffic_key == "bytes_per_call" else None)
                src = j.get("source")
                sha_ok = j.get("kernel_source_sha") == spmm_kernel_sha()
            except Exception:
                traffic = None
        return {"bound": "hbm", "kernel": f"spmm_csr_f32 d={d}", "achieved": round(ach, 2), "peak": HBM_PEAK_GBPS,
                "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBPS, 5),
                "frac_of_achievable_6300": round(ach / HBM_ACHIEVABLE_GBPS, 5),
                # HBM-side bytes per CALL from rocprofv3 --pmc passes of an earlier run of the same kernels (this run
                # measures time only): (2 * FETCH_SIZE + WRITE_SIZE) * 1024 summed over the call's launches
                "traffic": traffic, "traffic_source": src,
                "traffic_kernel_source_current": sha_ok,      # counters collected with the kernel sources of this tree?
                "ms_per_call": round(avg, 4), "ms_per_call_median": round(med, 4), "ms_per_call_min": round(mn, 4),
                "kernel_launches_per_call": launches_per_call, "calls_timed": int(a.size), "bytes_alg": b_alg,
                # what actually bounds the kernel: every gathered row crosses the L2 -> vector-L1 fill
                # path, 64 B/clk/CU x 256 CUs at the 2.345 GHz the chip holds under this load = 38.4 TB/s
                "gather_GBps": round(spmm_bytes_gather(nr, nz, d) / (avg * 1e-3) / 1e9, 1),
                "l1_fill_peak_GBps": L1_FILL_PEAK_GBPS,
                "l1_fill_frac": round(spmm_bytes_gather(nr, nz, d) / (avg * 1e-3) / 1e9 / L1_FILL_PEAK_GBPS, 4)}

    def launches(d):
        """average kernel launches of one SpMM call at width d over the calls of an epoch (forward and backward
        matrices are cut differently); single GPU only"""
        if multi:
            return None
        per = []
        for layer in G.layers():
            if min(layer.lin.W.n(), layer.lin.W.m()) != d:
                continue
            for buf in (layer.A.ext_buffer, layer.A.ext_buffer2):
                if buf is not None:
                    per.append(buf.num_launches(d))
        return round(float(np.mean(per)), 2) if per else None

    out = {
        "metric": "epoch_ms (Reddit-shaped 3x128 GCN, full-graph, fp32)" if args.workload == "reddit_like"
                  else "epoch_ms (ogbn-products-shaped 3x128 GCN, full-graph, fp32; manual mode)",
        "value": round(ms, 4), "unit": "ms", "n_gpus": P, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms, 4), "higher_is_better": False, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": (wl_name if args.scale == 1.0 else f"{args.workload}_scale_{args.scale}") + ("_symmetric" if args.symmetric else ""),
                   "n": n, "nnz": nnz, "sizes": sizes, "spmm_per_epoch": 2 * nl - 1,
                   "parallelism": f"rows{P}" + (f"-{args.mode}" if multi else ""), "fused": fused},
        "roofline": spmm_roofline(spmm_ms, d_main, launches(d_main), "bytes_per_call"),
        "roofline_narrow": spmm_roofline(narrow_ms, d_narrow, launches(d_narrow), "bytes_per_call_narrow") if narrow_timers else None,
        "loss_first_last": [round(float(losses[0]), 5), round(float(losses[-1]), 5)] if losses else None,
        "setup_s": round(t_gen, 1),
    }

    if multi and rank == 0 and os.environ.get("MGGCN_BENCH_HEADLINE_FILE"):
        # the headline is measured: park it where the launching process finds it, should one of the extras below (more
        # collectives, the other piece count) take the ranks down before the line is printed
        try:
            with open(os.environ["MGGCN_BENCH_HEADLINE_FILE"], "w") as fh:
                json.dump(out, fh)
        except OSError:
            pass
    if multi:
        if os.environ.get("MGGCN_BENCH_FAIL_RANKS") == "2":     # test hook: the ranks die after the timed epochs
            dist.barrier()
            if rank == 0:
                shutil.rmtree(tmp, ignore_errors=True)
            sys.exit(3)
        out["comm"] = comm_report(torch, dist, dctx, G, epoch, rehearsal, local_rank, spmm_timers, args)
        out["comm"]["high_priority_stream"] = high_priority_comm
        out["comm"]["chunks"] = out["comm"].get("chunks") or D.default_chunks(P)
        if P > 1 and args.mode == "allgather" and not args.no_extras:
            # the same epoch with the OTHER piece count (2 <-> 4): how many pieces pay depends on what the links deliver against
            # what a piece costs the compute stream (model: profiles/experiments/rank_epoch_model_r04.log) -- one more data point
            # per N from the same run; never the headline.  Every rank takes part (the ranks re-load their rows: collectives).
            alt = 2 if D.default_chunks(P) != 2 else 4
            G = None
            Ad2, A_Td2, Xd2, Yd2, _ = D.load_rank_local(dctx, tmp, chunks=alt)
            G2 = D.dist_gcn(dctx, Ad2, A_Td2, sizes, fused=fused, mode=args.mode)
            for _ in range(2):
                G2.train_step(dctx, Xd2, Yd2, 1e-2, 0.9, 0.999, 5e-4, 1e-8)
            barrier()
            t_a = time.perf_counter()
            for _ in range(args.steps):
                G2.train_step(dctx, Xd2, Yd2, 1e-2, 0.9, 0.999, 5e-4, 1e-8)
            barrier()
            ms_a = (time.perf_counter() - t_a) * 1e3 / max(args.steps, 1)
            t = torch.tensor([ms_a], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            out["alt_chunks"] = {"chunks": alt, "epoch_ms": round(float(t.item()), 4)}
            G2 = Ad2 = A_Td2 = Xd2 = Yd2 = None
    if rank == 0 and not multi and not args.no_extras:
        # (1) the reference's own interface on the same workload: never the headline (it synchronises inside the loss
        #     like the reference, src/gcn.hpp:816; `value` goes through gcn.train_step with one sync per epoch)
        cli = run_cli_epochs(pkg, indptr, indices, data, X, Y, args.hidden)
        if cli:
            out.update(cli)
        # (1b) optional mode: layer 0's loop-invariant aggregation A_fwd X computed once (gcn.set_hoist_first_aggregation):
        #     6 SpMMs per epoch instead of the reference's 7 (src/gcn.hpp:437-446) -- reported separately, never `value`
        if args.workload == "reddit_like":
            G.set_hoist_first_aggregation(True)
            for _ in range(max(args.warmup, 1)):
                epoch()
            torch.cuda.synchronize()
            t_h = time.perf_counter()
            for _ in range(args.steps):
                lh = epoch()[0]
            torch.cuda.synchronize()
            out["hoisted_first_aggregation"] = {"epoch_ms": round((time.perf_counter() - t_h) * 1e3 / max(args.steps, 1), 4),
                                                "spmm_per_epoch": 2 * nl - 2, "loss_last": round(float(lh), 5),
                                                "note": "A_fwd.X precomputed once; not the reference's epoch"}
            G.set_hoist_first_aggregation(False)
        # (2) the stand-in with the real dataset's STRUCTURE (A = A^T: both matrices have power-law rows and popular
        #     columns, rows sorted): same model, same kernels, extra key only -- SURVEY.md 8(d) defined the headline graph
        if not args.symmetric and args.workload == "reddit_like":
            G = A = None                          # release the headline model's plans and buffers
            torch.cuda.empty_cache()
            t_sym = time.time()
            (ip2, ix2, dv2), X2, Y2 = pkg.datasets.synth_reddit_like(args.scale, seed=1, symmetric=True)
            G2 = pkg.gcn(pkg.csr_matrix(ip2, ix2, dv2, int(ip2.shape[0] - 1)), [X2.shape[1]] + list(args.hidden) + [1 + int(Y2.max())], fused=fused)
            X2d, Y2d = pkg.dn_matrix.from_numpy(X2), pkg.dn_matrix.from_numpy(Y2)
            for _ in range(args.warmup):
                G2.train_step(ctx, X2d, Y2d, 1e-2, 0.9, 0.999, 5e-4, 1e-8)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            per2 = []
            for _ in range(args.steps):
                G2.train_step(ctx, X2d, Y2d, 1e-2, 0.9, 0.999, 5e-4, 1e-8)
                per2.append([ctx.measure(t) for t in spmm_timers])
            torch.cuda.synchronize()
            out["symmetric_epoch_ms"] = round((time.perf_counter() - t2) * 1e3 / max(args.steps, 1), 4)
            out["symmetric_spmm_ms_per_call"] = round(float(np.mean(per2)), 4) if per2 and per2[0] else None
            out["symmetric_setup_s"] = round(time.time() - t_sym - (time.perf_counter() - t2), 1)
            del G2, X2d, Y2d, ip2, ix2, dv2, X2, Y2
    if rank == 0 and not multi and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(pkg, indptr, indices, data, n, X, Y, sizes, d_main, args.workload)
        if args.warmup + args.steps > 0 and "loss" in out["cpu_baseline"]:
            out["cpu_baseline"]["loss_matches_gpu_first"] = bool(
                abs(out["cpu_baseline"]["loss"] - losses[0]) <= 1e-4 * abs(out["cpu_baseline"]["loss"]))

    if multi:
        # the ranks let go of the GPUs (model, plans, shards), then rank 0 alone runs the drop-in CLI on the same files
        G = Ad = A_Td = Xd = Yd = None
        epoch = None
        import gc
        gc.collect()
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        dist.barrier()
        dist.destroy_process_group()
        if rank == 0:
            try:
                if not args.no_extras and args.workload == "reddit_like":
                    cli = run_cli_multi_gpu(tmp, P, args.hidden, args.mode, not args.no_overlap, rehearsal)
                    if cli:
                        out.update(cli)
            finally:
                shutil.rmtree(tmp, ignore_errors=True)
    if rank == 0:
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())


def dist_min_is_negative(torch, dist, ex, rehearsal):
    """did ANY rank fail its exchange-timing pass?  (one MIN all-reduce, executed by every rank)"""
    flag = torch.tensor([1.0 if ex else -1.0], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return float(flag[0]) < 0


def comm_report(torch, dist, dctx, G, epoch, rehearsal, local_rank, spmm_timers, args):
    """N > 1: what the run exchanged over -- backend, communicator size, every rank's device, the RCCL version -- and
    one SEPARATE pass of 3 epochs (after the timed ones) with the exchange events on: per d-wide SpMM the comm-stream
    time of its collectives (`exchange_ms`), the time the compute stream stalled waiting for pieces (`exposed_ms`)
    and overlap_frac = 1 - exposed / exchange.  Reference events: src/cuda_utils.hpp:61-89."""
    prop = torch.cuda.get_device_properties(local_rank)
    me = {"rank": dctx.rank, "device": local_rank, "name": prop.name, "arch": getattr(prop, "gcnArchName", None),
          "pci_bus_id": getattr(prop, "pci_bus_id", None), "uuid": str(getattr(prop, "uuid", ""))[:13]}
    devices = [None] * dctx.P
    dist.all_gather_object(devices, me)
    try:
        ver = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception:
        ver = None
    rep = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "devices": devices, "rccl_version": ver,
           "mode": args.mode, "overlap": not args.no_overlap, "rehearsal_gloo_on_one_gpu": bool(rehearsal),
           "chunks": int(os.environ.get("MGGCN_DIST_CHUNKS", "0")) or None}
    ex, wt = [], []
    # The epochs themselves run OUTSIDE any try: they are collectives, and a rank that swallowed an error in the middle of one
    # would leave its peers blocked in it -- an error there ends the run, like in the timed epochs.  Only the
    # rank-local read-out of the event timers is guarded: a failure there costs the comm figures, not the line.
    dctx.profile_exchange = True
    for it in range(3):
        epoch()
        if it == 0 or "exchange_error" in rep:
            continue                                              # first pass creates the events; after a read-out failure the
                                                                  # epochs still run (every rank makes the same collective calls)
        try:
            for t in spmm_timers:                                 # "<layer>_<0|1>_matmul-spmm"
                base = t[: -len("matmul-spmm")]
                names = [k for k in dctx.ctx.timers if k.startswith(base)]
                ex.append(sum(dctx.measure(k) for k in names if k.endswith("matmul-exchange")))
                wt.append(sum(dctx.measure(k) for k in names if k.endswith("matmul-bcast-wait")))
        except Exception as e:            # noqa: BLE001 -- reported in the line
            rep["exchange_error"] = repr(e)[:200]
            ex = []
    dctx.profile_exchange = False
    # every rank takes part in this reduction whatever happened above (a rank whose pass failed is seen by the MIN reduction below)
    e, w = (float(np.mean(ex)), float(np.mean(wt))) if ex else (0.0, 0.0)
    both = torch.tensor([e, w], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
    dist.all_reduce(both, op=dist.ReduceOp.MAX)
    failed = dist_min_is_negative(torch, dist, ex, rehearsal)
    e, w = float(both[0]), float(both[1])
    if not failed and e > 0:
        rep.update({"exchange_ms": round(e, 4), "exposed_ms": round(w, 4), "overlap_frac": round(1.0 - w / e, 4),
                    "exchange_note": "per d-wide SpMM, max over ranks, separate pass after the timed epochs"})
    return rep


def cpu_baseline(pkg, indptr, indices, data, n, X, Y, sizes, d, workload="reddit_like"):
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
    ts = []
    for _ in range(3):                           # median + min
        t = time.perf_counter(); orc.spmm(A_T, B); ts.append(time.perf_counter() - t)
    t_spmm = float(np.median(ts))
    nl = len(sizes) - 1
    projected = t_spmm * (2 * nl - 1) * 1.6      # SpMMs dominate; GEMM/elementwise ~ +60 % on CPU
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    res = {"unit": "ms", "cores": cores, "cpu_model": model, "kind": "port",
           "spmm_ms": round(t_spmm * 1e3, 1), "spmm_ms_min": round(min(ts) * 1e3, 1),
           "spmm_GBps_alg": round(spmm_bytes_alg(n, n, A.nnz, d) / t_spmm / 1e9, 2)}
    if projected <= 45.0:
        O = orc.Gcn(orc.Csr(indptr, indices, data, n), sizes)
        t = time.perf_counter()
        loss0, _ = O.train_forward(X, Y); O.backward(); O.adam_update()
        res["value"] = round((time.perf_counter() - t) * 1e3, 1)
        res["loss"] = round(float(loss0), 5)     # epoch-0 loss of the oracle: compare with loss_first (same weights, same inputs)
        res["sample"] = "1 full epoch of the same workload (oracle.Gcn, fp32, OpenMP)"
    else:
        frac = max(0.02, 20.0 / projected)
        gen = pkg.datasets.synth_products_like if workload == "products_like" else pkg.datasets.synth_reddit_like
        (ip, ix, dv), Xs, Ys = gen(frac)
        O = orc.Gcn(orc.Csr(ip, ix, dv, ip.shape[0] - 1), sizes)
        t = time.perf_counter()
        O.train_forward(Xs, Ys); O.backward(); O.adam_update()
        res["value"] = round((time.perf_counter() - t) * 1e3 / frac, 1)
        res["sample"] = f"1 epoch on a {frac:.3f}-scale graph (same mean degree), time divided by the scale"
    return res


if __name__ == "__main__":
    main()
