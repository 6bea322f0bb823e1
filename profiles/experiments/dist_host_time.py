"""Host cost of one epoch of the DISTRIBUTED path: dist_gcn over RCCL with ONE rank on a graph so small that the
GPU work is negligible -- the epoch time is then what the Python layer + ctypes + torch.distributed need to enqueue an
epoch.  At P = 8 a rank's GPU work on the Reddit shape is ~2.1 ms + the exchange, so this has to stay well below that.
Usage: python profiles/experiments/dist_host_time.py   (one GPU)"""
import os, sys, time, tempfile, shutil
import numpy as np
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.distributed as dist
import __graft_entry__ as g
pkg = g.load_package()
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
scale = float(os.environ.get("EXP_SCALE", "0.01"))
(ip, ix, dv), X, Y = pkg.datasets.synth_reddit_like(scale, seed=1)
n = ip.shape[0] - 1
sizes = [X.shape[1], 128, 128, 128, 1 + int(Y.max())]
med = lambda v: sorted(v)[len(v) // 2] * 1e3

ctx = pkg.context(0)
G = pkg.gcn(pkg.csr_matrix(ip, ix, dv, n), sizes, fused=True)
Xd, Yd = pkg.dn_matrix.from_numpy(X), pkg.dn_matrix.from_numpy(Y)
for _ in range(5): G.train_step(ctx, Xd, Yd, 1e-2, 0.9, 0.999, 5e-4, 1e-8)
ts = []
for _ in range(30):
    t0 = time.perf_counter(); G.train_step(ctx, Xd, Yd, 1e-2, 0.9, 0.999, 5e-4, 1e-8); ts.append(time.perf_counter() - t0)
print(f"n = {n}: single-GPU gcn epoch {med(ts):.3f} ms (host-bound at this size)", flush=True)

D = pkg.dist
tmp = tempfile.mkdtemp(prefix="mggcn_hosttime_")
pkg.datasets.write_dataset(tmp, ip, ix, dv, X, Y)
for mode in ("allgather", "halo", "rounds"):
    for overlap in (True, False):
        dctx = D.dist_context(overlap=overlap, device_index=0)
        Ad, A_Td, Xr, Yr, info = D.load_rank_local(dctx, tmp)
        Gd = D.dist_gcn(dctx, Ad, A_Td, sizes, fused=True, mode=mode)
        for _ in range(5): Gd.train_step(dctx, Xr, Yr, 1e-2, 0.9, 0.999, 5e-4, 1e-8)
        ts = []
        for _ in range(30):
            t0 = time.perf_counter(); Gd.train_step(dctx, Xr, Yr, 1e-2, 0.9, 0.999, 5e-4, 1e-8); ts.append(time.perf_counter() - t0)
        print(f"n = {n}: dist_gcn (1 rank, RCCL) mode={mode:9s} overlap={int(overlap)}: epoch {med(ts):.3f} ms", flush=True)
shutil.rmtree(tmp, ignore_errors=True)
dist.destroy_process_group()
