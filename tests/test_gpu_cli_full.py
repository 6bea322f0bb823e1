"""The north-star interface at the sizes BASELINE.json names (-m gpu):

  C2  `mg_gcn -E 6 train <reddit-shaped dir> 3 128 128 128`                    src/main.cpp:113-131, README.md:44
  C3  `mg_gcn -P 8 -R 1 -E 2 train ...` through dist_row_csr_matrix, the K-piece exchange (and the reference's
      rounds), libmggcn_comm and the fused [G_W | G_b] all-reduce AT ITS OWN SIZE                src/main.cpp:134-170
      -- on this one-GPU box the eight ranks wrap over the card (MGGCN_OVERSUBSCRIBE=1: the library's event-ordered
      peer-copy transport; RCCL refuses two ranks per device), so the schedule, the partition and every kernel run
      at the real shapes; only the wire is not xGMI.
  C3' the one-process-per-GPU Python form: 4 `gloo` ranks on the card (the box allows 6 GPU processes), every rank
      loading only its rows of the files (dist.load_rank_local), K-piece all-gather schedule.

The SURVEY.md 8(d) Reddit stand-in (tests/conftest.py; the symmetric one runs at full size in tests/test_gpu_configs.py).  Epoch-0 loss against
the CPU oracle (exact-accumulation twin, pinned in tests/test_oracle_kat.py) at 1e-4; for P > 1 the oracle runs with the class count
padded to a multiple of P (src/main.cpp:135) -- the row-partitioned model computes the same function, regrouped.
The CLI's own per-epoch seconds are printed (pytest -s) and bounded loosely; bench.py reports them as `cli_epoch_ms`."""
import os
import re
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "mg-gcn_amd", "bin", "mg_gcn")
TOL = 1e-4
HIDDEN = ["3", "128", "128", "128"]


def _run(args, cwd, env=None, timeout=900):
    e = dict(os.environ)
    e.update(env or {})
    return subprocess.run([BIN] + args, cwd=cwd, env=e, capture_output=True, text=True, timeout=timeout)


def _epoch_lines(stderr):
    """the 'e loss acc seconds' lines (src/main.cpp:130, :167)"""
    out = []
    for ln in stderr.splitlines():
        t = ln.split()
        if len(t) == 4 and t[0].isdigit():
            try:
                out.append((int(t[0]), float(t[1]), float(t[2]), float(t[3])))
            except ValueError:
                pass
    return out


def _check_header(stderr):
    lines = stderr.strip().splitlines()
    assert lines[0] == "232968 114848860", lines[:3]                     # test/test_matrix.cpp:48-58
    assert lines[1] == "num_labels = 41" and lines[2] == "feature size = 608", lines[:3]


def _check_loss(got, oracle, data, classes):
    """against the exact-accumulation twin (the fp32 restatement's own distance to it at this size is measured in
    tests/test_gpu_configs.py::test_c2_full_epoch_matches_oracle; one oracle epoch is ~6 s of host time)"""
    from conftest import reddit_oracle_epoch
    want = reddit_oracle_epoch(oracle, data, classes, True)
    assert abs(got[1] - want["loss"]) <= TOL * abs(want["loss"]), (data["kind"], classes, got, want["loss"])
    assert abs(got[2] - want["acc"]) <= 8.0 / data["n"], (got, want["acc"])


def test_cli_c2_full_reddit(pkg, oracle, reddit_dirs):
    # the SURVEY 8(d) stand-in.  (The symmetric one goes through the same kernels at full size in tests/test_gpu_configs.py;
    # the second 35 s of CLI runs on it that this file held bought nothing the suite does not already hold -- the suite's
    # wall time is a budget too.)
    from conftest import reddit_standin
    reddit_any = reddit_standin(pkg, "asym")
    d = reddit_dirs(reddit_any["kind"])
    r = _run(["-E", "6", "train", d] + HIDDEN, cwd=reddit_dirs.base)
    assert r.returncode == 0, r.stderr[-2000:]
    _check_header(r.stderr)
    ep = _epoch_lines(r.stderr)
    assert [e[0] for e in ep] == list(range(6))
    _check_loss(ep[0], oracle, reddit_any, 41)
    assert ep[-1][1] < ep[0][1]                                           # trains
    med = float(np.median([e[3] for e in ep[2:]])) * 1e3                  # epochs 2.. (0/1 carry set-up)
    print(f"\n[cli C2 {reddit_any['kind']}] median epoch {med:.2f} ms, first epoch (plans + lazy set-up) {ep[0][3]:.2f} s")
    assert med < 40.0                                                     # sanity only: 16-17 ms measured
    csv = os.path.join(reddit_dirs.base, "csvs", f"permuted_reddit_{reddit_any['kind']}_608_128_128_128_41_1.csv")
    text = open(csv).read()                                               # file name scheme src/main.cpp:100-111
    for name in ("0_0_matmul-spmm", "3_1_matmul-spmm", "4_loss-layer"):    # the reference's timer names
        assert re.search(rf"^5_0_{name}:", text, re.M), name


@pytest.mark.parametrize("kind,mode", [("asym", "allgather"), ("asym", "rounds")])
def test_cli_c3_p8_oversubscribed(pkg, oracle, reddit_dirs, kind, mode):
    from conftest import reddit_standin
    data = reddit_standin(pkg, kind)
    d = reddit_dirs(kind)
    r = _run(["-P", "8", "-R", "1", "-E", "2", "train", d] + HIDDEN, cwd=reddit_dirs.base,
             env={"MGGCN_OVERSUBSCRIBE": "1", "MGGCN_DIST_MODE": mode})
    assert r.returncode == 0, r.stderr[-2000:]
    _check_header(r.stderr)
    ep = _epoch_lines(r.stderr)
    assert [e[0] for e in ep] == [0, 1]
    _check_loss(ep[0], oracle, data, 48)                                  # 41 classes padded to 48 (src/main.cpp:135)
    assert ep[1][1] < ep[0][1]
    print(f"\n[cli C3 P=8 one GPU, {kind}, {mode}] epoch 1: {ep[1][3] * 1e3:.1f} ms; start-up + epoch 0: {ep[0][3]:.1f} s")
    text = open(os.path.join(reddit_dirs.base, "csvs", f"permuted_reddit_{kind}_608_128_128_128_41_8.csv")).read()
    assert re.search(r"^1_7_0_0_matmul-spmm:", text, re.M)                # per-rank timers "<epoch>_<rank>_<name>"


# ------------------------------------------------------------------------------------------------
# the one-process-per-GPU form at full size
# ------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _rank(rank, P, port, dirname, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=P)
    try:
        sys.path.insert(0, ROOT)
        import __graft_entry__ as ge
        pkg = ge.load_package()
        D = pkg.dist
        dctx = D.dist_context(overlap=True, device_index=0)
        Ad, A_Td, Xd, Yd, info = D.load_rank_local(dctx, dirname)          # only this rank's rows of the files
        sizes = [info["features"], 128, 128, 128, (info["num_labels"] + P - 1) // P * P]
        G = D.dist_gcn(dctx, Ad, A_Td, sizes, fused=True, mode="allgather")
        loss, acc = G.train_forward(dctx, Xd, Yd)
        G.backward(dctx)
        dctx.sync()
        grads = [(l.GW().local.numpy().copy(), l.Gb().local.numpy().copy()) for l in G.layers()] if rank == 0 else None
        loss1, _ = G.train_step(dctx, Xd, Yd, 1e-2, 0.9, 0.999, 5e-4, 1e-8)      # same parameters: the same forward again
        loss2, _ = G.train_step(dctx, Xd, Yd, 1e-2, 0.9, 0.999, 5e-4, 1e-8)      # after one Adam step
        q.put((rank, loss, acc, (loss1, loss2), grads, info["nnz_local"], info["host_bytes"]))
    finally:
        dist.destroy_process_group()


def test_python_ranks_c3_full_size(pkg, oracle, reddit_dirs):
    """4 ranks x 58 242 rows; loss and every all-reduced gradient of epoch 0 against the exact-accumulation oracle
    with 44 classes (41 padded to a multiple of 4)."""
    from conftest import reddit_oracle_epoch, reddit_standin
    P = 4
    data = reddit_standin(pkg, "asym")
    d = reddit_dirs("asym")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank, args=(r, P, port, d, q)) for r in range(P)]
    for pr in procs:
        pr.start()
    res = sorted([q.get(timeout=600) for _ in range(P)], key=lambda t: t[0])
    for pr in procs:
        pr.join(timeout=120)
        assert pr.exitcode == 0
    want = reddit_oracle_epoch(oracle, data, 44, True)
    assert sum(r[5] for r in res) == 114_848_860                          # the ranks' row blocks cover the graph
    assert max(r[5] for r in res) < 0.3 * 114_848_860                     # ... and no rank read more than its share
    for rank, loss, acc, (loss1, loss2), grads, _, _ in res:
        assert abs(loss - want["loss"]) <= TOL * abs(want["loss"]), (rank, loss, want["loss"])
        assert loss == res[0][1] and (loss1, loss2) == res[0][3]          # one global loss on every rank
        # deterministic forward (train_step sums the ranks' loss terms inside the last layer's gradient all-reduce, train_forward
        # in a collective of its own: the same four numbers added in a possibly different order); then it trains
        assert abs(loss1 - loss) <= 1e-6 * abs(loss) and loss2 < loss1
    for (gw, gb), (ow, ob) in zip(res[0][4], want["grads"]):
        assert np.abs(gw - ow).max() <= TOL * np.abs(ow).max()
        assert np.abs(gb - ob).max() <= TOL * np.abs(ob).max()
