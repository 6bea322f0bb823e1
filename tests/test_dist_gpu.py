"""The 1D-row-partition path with the REAL HIP kernels on more than one rank: two (and
three) processes share the box's single GPU and exchange through `gloo` (RCCL refuses two
ranks on one device; the all-gather / broadcast / all-reduce code paths above the transport
are the same).  Checked against the oracle's P-shard simulation and against the single-GPU
model with the same padded class count (src/main.cpp:135)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _data(n, F, C):
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    pkg = ge.load_package()
    ip, ix, dv = pkg.datasets.synth_powerlaw_csr(n, n * 16, 700, seed=21)
    rng = np.random.default_rng(22)
    X = rng.standard_normal((n, F), dtype=np.float32)
    Y = rng.integers(0, C, size=(n, 1)).astype(np.int32)
    return pkg, (ip, ix, dv), X, Y


def _worker(rank, P, port, n, F, C, hidden, mode, epochs, q, backend="gloo", chunks=None, overlap=True, resync=None):
    """resync[e] = the oracle's [(W, b) per layer] after ITS Adam step of epoch e.  Adam's first steps are
    lr * g / (|g| + eps): the sign of a rounding-noise gradient decides a whole 0.01 step, so two correct summation
    orders drift apart in free running.  Every rank therefore (1) checks its own updated parameters against the
    oracle's up to such sign flips (2 lr) and (2) continues from the oracle's -- so that EVERY epoch, not only epoch 0,
    is comparable at 1e-4 (the later epochs are the ones that re-use every buffer, event and exchange slot)."""
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    if backend == "nccl":
        import torch
        os.environ["MGGCN_DIST_SELF_GATHER"] = "1"          # one rank: the all-gather over ProcessGroupNCCL still runs
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=rank, world_size=P, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=P)
    try:
        pkg, (ip, ix, dv), X, Y = _data(n, F, C)
        D = pkg.dist
        dctx = D.dist_context(overlap=overlap, device_index=0)
        assert dctx.bcast_stream_id() == (1 if overlap else 0)            # src/dist_matrix.hpp:20-22
        A = pkg.csr_matrix(ip, ix, dv, n)
        A.normalize(True)
        A_T = A.transpose()
        p = D.partition_bounds(n, P)
        sizes = [F] + hidden + [(C + P - 1) // P * P]
        G = D.dist_gcn(dctx, D.dist_row_csr_matrix(dctx, A, p, p, chunks), D.dist_row_csr_matrix(dctx, A_T, p, p, chunks),
                       sizes, fused=True, mode=mode)
        Xd, Yd = D.dist_row_dn_matrix(dctx, X), D.dist_row_dn_matrix(dctx, Y)
        out = []
        for ep in range(epochs):
            if ep == epochs - 1 and epochs > 1:         # last epoch through the one-sync step
                loss, acc = G.train_step(dctx, Xd, Yd, 1e-2, 0.9, 0.999, 5e-4, 1e-8)
                out.append((loss, acc, None, None))
                continue
            loss, acc = G.train_forward(dctx, Xd, Yd)
            G.backward(dctx)
            dctx.sync()
            grads = [l.GW().local.numpy().copy() for l in G.layers()]
            gb = [l.Gb().local.numpy().copy() for l in G.layers()]
            G.adam_update(dctx, 1e-2, 0.9, 0.999, 5e-4, 1e-8)
            dctx.sync()
            out.append((loss, acc, grads, gb))
            if resync is not None:
                for l, (W, b) in zip(G.layers(), resync[ep]):
                    assert np.abs(l.W().local.numpy() - W).max() <= 2.05e-2      # (1)
                    assert np.abs(l.b().local.numpy() - b).max() <= 2.05e-2
                    l.W().local.init(W)                                           # (2)
                    l.b().local.init(b)
                dctx.sync()
        q.put((rank, out, [l.W().local.numpy() for l in G.layers()]))
    finally:
        dist.destroy_process_group()


def _oracle_epochs(O, X, Y, epochs):
    """the oracle's own run: per epoch (loss, acc, [G_W], [G_b]) and the parameters after its Adam step"""
    want, resync = [], []
    for _ in range(epochs):
        ol, oa = O.train_forward(X, Y)
        O.backward()
        want.append((ol, oa, [l.lin.G_W.copy() for l in O.ranks[0]], [l.lin.G_b.copy() for l in O.ranks[0]]))
        O.adam_update()
        resync.append([(l.lin.W.copy(), l.lin.b.copy()) for l in O.ranks[0]])
    return want, resync


def _assert_epochs_match(rank, out, want, n, model_floor=0.0):
    """model_floor: a gradient tensor is held to 1e-4 of max(its own largest entry, model_floor x the model's largest
    gradient entry) -- 0 for the regular shapes; the odd-shape cases (width-1 bottlenecks) have layers whose whole gradient
    is rounding noise next to the others"""
    for e, ((loss, acc, grads, gb), (ol, oa, oG, oGb)) in enumerate(zip(out, want)):
        assert abs(loss - ol) <= 1e-4 * abs(ol), (rank, e, loss, ol)       # EVERY epoch at the north-star bar
        assert abs(acc - oa) <= 3.0 / n, (rank, e, acc, oa)
        if grads is None:
            continue
        gmax = max(np.abs(og).max() for og in list(oG) + list(oGb))
        for g, og in zip(grads, oG):                                        # all-reduced gradients, every rank
            assert np.abs(g - og).max() <= 1e-4 * max(np.abs(og).max(), model_floor * gmax), (rank, e)
        for g, og in zip(gb, oGb):
            assert np.abs(g - og).max() <= 1e-4 * max(np.abs(og).max(), model_floor * gmax), (rank, e)


@pytest.mark.parametrize("P,mode,chunks,overlap", [
    (2, "allgather", None, True), (2, "rounds", None, True), (3, "allgather", None, True), (2, "allgather", 3, True),
    (4, "allgather", 2, True), (2, "halo", None, True), (3, "halo", None, True),
    # the reference's -S flag: overlap = false puts the exchange on the compute stream (src/dist_matrix.hpp:20-22,
    # src/main.cpp:66) -- same results, every schedule
    (2, "allgather", None, False), (2, "rounds", None, False), (2, "halo", None, False)])
def test_dist_gcn_matches_oracle(oracle, P, mode, chunks, overlap):
    _dist_case(oracle, P, mode, chunks, overlap, 1536, 20, 5, [16, 16], 3)


# corners nobody chose: shards of 8 rows, no hidden layer, width-1 layers, class counts padded to a multiple of P, more pieces
# than a shard has rows, five ranks (the box allows six GPU processes)
@pytest.mark.parametrize("P,mode,chunks,overlap,n,F,C,hidden", [
    (2, "allgather", None, True, 16, 3, 2, []), (3, "halo", None, True, 393, 1, 7, [5]), (4, "rounds", None, True, 64, 33, 5, [1, 40]),
    (2, "allgather", 5, False, 3000, 7, 2, [200, 1, 64]), (5, "allgather", None, True, 1000, 100, 4, [16]),
    (4, "halo", None, False, 32, 5, 9, [8]), (2, "allgather", 64, True, 16, 2, 3, [4])])
def test_dist_gcn_on_odd_shapes(oracle, P, mode, chunks, overlap, n, F, C, hidden):
    _dist_case(oracle, P, mode, chunks, overlap, n, F, C, hidden, 3, must_train=False, model_floor=1e-2)


def _dist_case(oracle, P, mode, chunks, overlap, n, F, C, hidden, epochs, must_train=True, model_floor=0.0):
    _, (ip, ix, dv), X, Y = _data(n, F, C)
    O = oracle.DistGcn(oracle.Csr(ip, ix, dv, n), [F] + hidden + [C], P)
    want, resync = _oracle_epochs(O, X, Y, epochs)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, P, port, n, F, C, hidden, mode, epochs, q, "gloo", chunks, overlap, resync))
             for r in range(P)]
    for pr in procs:
        pr.start()
    res = sorted([q.get(timeout=300) for _ in range(P)], key=lambda t: t[0])
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    for rank, out, W in res:
        _assert_epochs_match(rank, out, want, n, model_floor)
        assert out[-1][0] < out[0][0] or not must_train                     # trains
    # replicated weights stay bitwise identical across ranks (same all-reduced gradient, same Adam)
    for li in range(len(res[0][2])):
        for r in range(1, P):
            np.testing.assert_array_equal(res[0][2][li], res[r][2][li])
    for r in range(1, P):
        for e in range(epochs):
            assert res[r][1][e][0] == res[0][1][e][0]                       # same global loss on every rank


@pytest.mark.parametrize("overlap", [True, False])
@pytest.mark.parametrize("mode", ["allgather", "rounds", "halo"])
def test_dist_gcn_over_rccl_single_rank(oracle, mode, overlap):
    """The RCCL transport itself (backend "nccl": all_gather_into_tensor / broadcast / all_reduce on
    the comm stream, stream-level waits) with the one rank a one-GPU box allows; the multi-rank
    logic above it is what the gloo cases check."""
    n, F, C, hidden, epochs = 1536, 20, 5, [16, 16], 3
    _, (ip, ix, dv), X, Y = _data(n, F, C)
    O = oracle.DistGcn(oracle.Csr(ip, ix, dv, n), [F] + hidden + [C], 1)
    want, resync = _oracle_epochs(O, X, Y, epochs)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    pr = ctx.Process(target=_worker, args=(0, 1, _free_port(), n, F, C, hidden, mode, epochs, q, "nccl", 2, overlap, resync))
    pr.start()
    rank, out, W = q.get(timeout=300)
    pr.join(timeout=60)
    assert pr.exitcode == 0
    _assert_epochs_match(0, out, want, n)
    assert out[-1][0] < out[0][0]


def _partitioned_worker(rank, P, port, dirname, hidden, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=P)
    try:
        sys.path.insert(0, ROOT)
        import __graft_entry__ as ge
        pkg = ge.load_package()
        D = pkg.dist
        dctx = D.dist_context(overlap=True, device_index=0)
        out = {}
        for mode in ("halo", "allgather"):
            Ad, A_Td, Xd, Yd, info = D.load_rank_local(dctx, dirname)           # rank-local: only this rank's rows
            sizes = [info["features"]] + hidden + [(info["num_labels"] + P - 1) // P * P]
            G = D.dist_gcn(dctx, Ad, A_Td, sizes, fused=True, mode=mode)
            loss, acc = G.train_forward(dctx, Xd, Yd)
            G.backward(dctx)
            dctx.sync()
            out[mode] = (loss, acc, [l.GW().local.numpy().copy() for l in G.layers()])
            if mode == "halo":
                out["halo_rows"] = sum(A_Td.halo["recv_rows"]) + sum(Ad.halo["recv_rows"])
        q.put((rank, out, info["n"]))
    finally:
        dist.destroy_process_group()


def test_halo_on_a_partitioned_graph_moves_less_and_matches_allgather(pkg, oracle, tmp_path):
    """The reason mode="halo" exists: on a community graph cut by the in-repo partitioner
    (datasets.partition_blocks through prepare_dataset's hook) the halo exchange moves well under half of what the
    all-gather moves, and trains to the same numbers.  Every rank loads only its own rows (dist.load_rank_local)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_dist_cpu import _community_graph
    P, n, hidden = 2, 2048, [16, 16]
    A = _community_graph(n, P, 14, 0.95, seed=5)
    rng = np.random.default_rng(6)
    X = rng.standard_normal((n, 12)).astype(np.float32)
    Y = rng.integers(0, 5, size=n)
    d = pkg.datasets.prepare_dataset(str(tmp_path / "comm"), A, X, Y, P=P, partitioner="blocks")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_partitioned_worker, args=(r, P, port, d, hidden, q)) for r in range(P)]
    for pr in procs:
        pr.start()
    res = sorted([q.get(timeout=300) for _ in range(P)], key=lambda t: t[0])
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    (ip, ix, dv, nn, _), Xf, Yf, _ = pkg.datasets.read_dataset(d)
    O = oracle.DistGcn(oracle.Csr(ip, ix, dv, nn), [Xf.shape[1]] + hidden + [1 + int(Yf.max())], P)
    ol, oa = O.train_forward(Xf, Yf)
    O.backward()
    halo_rows = 0
    for rank, out, n_ in res:
        for mode in ("halo", "allgather"):
            loss, acc, grads = out[mode]
            assert abs(loss - ol) <= 1e-4 * abs(ol), (rank, mode, loss, ol)
            for g, l in zip(grads, O.ranks[0]):
                assert np.abs(g - l.lin.G_W).max() <= 1e-4 * np.abs(l.lin.G_W).max(), (rank, mode)
        for gh, ga in zip(out["halo"][2], out["allgather"][2]):
            assert np.abs(gh - ga).max() <= 1e-4 * np.abs(ga).max()
        halo_rows += out["halo_rows"]
    # both matrices (forward + backward) over all ranks vs the all-gather's (P - 1) n rows per matrix
    assert halo_rows < 0.5 * 2 * (P - 1) * nn, (halo_rows, 2 * (P - 1) * nn)
