"""Host time of one epoch of the Python form as a rank of an 8-GPU job issues it: dist_gcn over RCCL with ONE rank but the exchange
of P = 8 (MGGCN_DIST_SELF_GATHER=1, 4 pieces per SpMM: 28 all-gathers + 4 all-reduces per epoch through ProcessGroupNCCL), on a
graph so small that the device does next to nothing.  `issue` = wall time until everything is enqueued (before the epoch's one
synchronisation), `epoch` = with the synchronisation and the loss read.  A rank's device work at P = 8 is ~3.1 ms + the exchange."""
import os, sys, time, tempfile, shutil
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29534")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
os.environ["MGGCN_DIST_SELF_GATHER"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.distributed as dist
import __graft_entry__ as g
pkg = g.load_package()
torch.cuda.set_device(0)
opts = dist.ProcessGroupNCCL.Options(); opts.is_high_priority_stream = True
dist.init_process_group("nccl", device_id=torch.device("cuda", 0), pg_options=opts)
(ip, ix, dv), X, Y = pkg.datasets.synth_reddit_like(0.01, seed=1)
n = ip.shape[0] - 1
sizes = [X.shape[1], 128, 128, 128, 48]
med = lambda v: sorted(v)[len(v) // 2] * 1e3
D = pkg.dist
tmp = tempfile.mkdtemp(prefix="mggcn_hosttime_")
pkg.datasets.write_dataset(tmp, ip, ix, dv, X, Y)
for chunks in [int(x) for x in os.environ.get("EXP_CHUNKS", "1,2,4").split(",")]:
    dctx = D.dist_context(overlap=True, device_index=0)
    Ad, A_Td, Xr, Yr, info = D.load_rank_local(dctx, tmp, chunks=chunks)
    Gd = D.dist_gcn(dctx, Ad, A_Td, sizes, fused=True, mode="allgather")
    for _ in range(5): Gd.train_step(dctx, Xr, Yr, 1e-2, 0.9, 0.999, 5e-4, 1e-8)
    issue, total, syn, rd = [], [], [], []
    for _ in range(40):
        t0 = time.perf_counter()
        out = Gd(dctx, Xr); Gd.loss_layer(dctx, out, Yr, sync=False); Gd.backward(dctx); Gd.adam_update(dctx, 1e-2, 0.9, 0.999, 5e-4, 1e-8)
        t1 = time.perf_counter()
        dctx.sync()
        t2 = time.perf_counter()
        Gd.loss_layer.read(dctx)
        t3 = time.perf_counter()
        issue.append(t1 - t0); total.append(t3 - t0); syn.append(t2 - t1); rd.append(t3 - t2)
    print(f"n = {n}, {chunks} piece(s) per SpMM ({7 * chunks} all-gathers + 4 all-reduces per epoch): issue {med(issue):.3f} ms, sync {med(syn):.3f} ms, loss read {med(rd):.3f} ms, epoch {med(total):.3f} ms", flush=True)
shutil.rmtree(tmp, ignore_errors=True)
dist.destroy_process_group()
