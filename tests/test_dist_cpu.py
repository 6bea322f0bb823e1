"""World-size-2 (and 4) `gloo` rehearsal of the 1D-row-partition path on CPU: the host-side
partitioning (block split, diagonal/remote split) and the exchange step (all-gather, the
reference's broadcast rounds, gradient all-reduce) run for real through torch.distributed;
the arithmetic between them is the oracle's SpMM (the HIP kernels need a GPU -- their
multi-rank test is tests/test_dist_gpu.py).  Checked against the single-process result."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _graph(n, seed):
    import scipy.sparse as sp
    M = sp.random(n, n, density=0.08, format="csr", dtype=np.float32, random_state=seed)
    M = sp.csr_matrix(M + sp.eye(n, dtype=np.float32, format="csr"))
    M.data = (M.data + 0.25).astype(np.float32)
    return M.indptr.astype(np.uint32), M.indices.astype(np.uint32), M.data


def _worker(rank, P, port, n, d, out_q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import __graft_entry__ as ge
    import oracle as orc
    pkg = ge.load_package()
    D = pkg.dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=P)
    try:
        ip, ix, dv = _graph(n, 3)
        A = pkg.csr_matrix(ip, ix, dv, n)
        A.normalize(True)
        p = D.partition_bounds(n, P)
        blocks = D.split_row_block(A, p[rank], p[rank + 1], p)
        diag, remote = D.split_local_remote(A, p[rank], p[rank + 1])
        rng = np.random.default_rng(11)
        B = rng.standard_normal((n, d)).astype(np.float32)       # same on every rank
        mine = torch.from_numpy(B[p[rank]:p[rank + 1]].copy())
        as_o = lambda m: orc.Csr(m.indptr, m.indices, m.data, m.m())

        # all-gather schedule: local block first, then the merged remote blocks (beta = 1)
        gathered = D.gloo_all_gather_rows(mine, P).numpy()
        np.testing.assert_array_equal(gathered, B)
        C = orc.spmm(as_o(diag), mine.numpy())
        orc.spmm(as_o(remote), gathered, C, 1.0, 1.0)

        # the same exchange cut into K pieces of every shard (K all-gathers; remote block cut by piece)
        rows = p[rank + 1] - p[rank]
        Cc = {}
        for K in (2, 3):
            cb = D.chunk_bounds(rows, K)
            acc = orc.spmm(as_o(diag), mine.numpy())
            for c, blk in enumerate(D.split_remote_chunks(remote, P, rows, K)):
                piece = D.gloo_all_gather_rows(mine[cb[c]:cb[c + 1]].contiguous(), P).numpy()
                assert piece.shape == (P * (cb[c + 1] - cb[c]), d) and blk.m() == piece.shape[0]
                orc.spmm(as_o(blk), piece, acc, 1.0, 1.0)
            Cc[K] = acc

        # halo exchange: every peer gets only the rows of this shard its blocks reference
        need = D.halo_need_lists(blocks, rank)
        send = [np.unique(D.split_row_block(A, p[s], p[s + 1], p)[rank].indices) if s != rank else np.empty(0, np.int64)
                for s in range(P)]
        sendbuf = torch.cat([mine[torch.from_numpy(np.asarray(x, dtype=np.int64))] for x in send]) if P > 1 else mine[:0]
        recvbuf = torch.empty((sum(len(x) for x in need), d), dtype=torch.float32)
        dist.all_to_all_single(recvbuf, sendbuf.contiguous(), [len(x) for x in need], [len(x) for x in send])
        Hh = orc.spmm(as_o(diag), mine.numpy())
        merged = D.merge_blocks_halo(blocks, need, rank)
        assert merged.m() == recvbuf.shape[0] and merged.nnz() == remote.nnz()
        orc.spmm(as_o(merged), recvbuf.numpy(), Hh, 1.0, 1.0)
        halo_rows = recvbuf.shape[0]

        # the reference's rounds: broadcast shard i, multiply block (rank, i), accumulate in order
        R = np.empty_like(C)
        for i in range(P):
            shard = D.gloo_broadcast_rows(mine if rank == i else None, (p[i + 1] - p[i], d), torch.float32, i, rank)
            orc.spmm(as_o(blocks[i]), shard.numpy(), R, 1.0, 0.0 if i == 0 else 1.0)

        # weight-gradient all-reduce: G_W = X^T G summed over ranks == full product
        G = rng.standard_normal((n, 5)).astype(np.float32)
        gw = torch.from_numpy(orc.gemm(B[p[rank]:p[rank + 1]], G[p[rank]:p[rank + 1]], A_T=True))
        gw = D.gloo_all_reduce_sum(gw.reshape(-1)).reshape(d, 5).numpy()
        out_q.put((rank, C, R, gw, Cc, Hh, halo_rows))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("P", [2, 4])
def test_partition_and_exchange_match_single_process(oracle, pkg, P):
    n, d = 64, 16
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, P, port, n, d, q)) for r in range(P)]
    for pr in procs:
        pr.start()
    res = [q.get(timeout=120) for _ in range(P)]
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    res.sort(key=lambda t: t[0])

    ip, ix, dv = _graph(n, 3)
    A = oracle.Csr(ip, ix, dv, n)
    oracle.normalize(A, True)
    B = np.random.default_rng(11).standard_normal((n, d)).astype(np.float32)
    full = oracle.spmm(A, B, f64acc=True)
    p = [i * n // P for i in range(P + 1)]
    blocks = oracle.block_split(A, p, p)
    rounds = oracle.dist_spmm(blocks, [B[p[j]:p[j + 1]] for j in range(P)])
    G = np.random.default_rng(11)
    G.standard_normal((n, d))
    Gm = G.standard_normal((n, 5)).astype(np.float32)
    gw_full = oracle.gemm(B, Gm, A_T=True, f64acc=True)
    for rank, C, R, gw, Cc, Hh, halo_rows in res:
        np.testing.assert_allclose(C, full[p[rank]:p[rank + 1]], rtol=1e-5, atol=1e-6)     # regrouped sum
        np.testing.assert_array_equal(Hh, C)           # halo: same products in the same order as the all-gather form
        assert 0 < halo_rows <= n - n // P
        for K in (2, 3):
            np.testing.assert_allclose(Cc[K], full[p[rank]:p[rank + 1]], rtol=1e-5, atol=1e-6)
        np.testing.assert_array_equal(R, rounds[rank])                                      # same order: bit-exact
        np.testing.assert_allclose(gw, gw_full, rtol=1e-5, atol=1e-5)


def test_local_remote_split_reassembles(pkg, oracle):
    D = pkg.dist
    n = 48
    ip, ix, dv = _graph(n, 7)
    A = pkg.csr_matrix(ip, ix, dv, n)
    dense = A.as_dn()
    for P in (2, 3, 4):
        p = D.partition_bounds(n, P)
        for r in range(P):
            diag, remote = D.split_local_remote(A, p[r], p[r + 1])
            assert (diag.n(), diag.m()) == (n // P, n // P) and (remote.n(), remote.m()) == (n // P, n)
            want = dense[p[r]:p[r + 1]].copy()
            np.testing.assert_array_equal(diag.as_dn(), want[:, p[r]:p[r + 1]])
            want[:, p[r]:p[r + 1]] = 0
            np.testing.assert_array_equal(remote.as_dn(), want)
            assert diag.nnz() + remote.nnz() == int(A.indptr[p[r + 1]] - A.indptr[p[r]])
    with pytest.raises(ValueError):
        D.partition_bounds(10, 4)            # n % P != 0 (reference asserts, dist_matrix.hpp:428)


def test_remote_chunks_reassemble(pkg):
    """split_remote_chunks: piece c of the remote block, with the columns renumbered to the layout
    an all-gather of piece c of every shard produces, holds exactly the entries of those columns."""
    D = pkg.dist
    n = 60
    ip, ix, dv = _graph(n, 9)
    A = pkg.csr_matrix(ip, ix, dv, n)
    for P in (2, 3, 4):
        p = D.partition_bounds(n, P)
        rows = n // P
        for r in range(P):
            _, remote = D.split_local_remote(A, p[r], p[r + 1])
            dense = remote.as_dn()
            for K in (1, 2, 4, 7, rows):
                cb = D.chunk_bounds(rows, K)
                chunks = D.split_remote_chunks(remote, P, rows, K)
                assert len(chunks) == K and sum(c.nnz() for c in chunks) == remote.nnz()
                for c, blk in enumerate(chunks):
                    ln = cb[c + 1] - cb[c]
                    assert (blk.n(), blk.m()) == (rows, P * ln)
                    cols = np.concatenate([np.arange(s * rows + cb[c], s * rows + cb[c + 1]) for s in range(P)])
                    np.testing.assert_array_equal(blk.as_dn(), dense[:, cols])
    assert D.default_chunks(1) == 1 and D.default_chunks(2) == 2 and D.default_chunks(8) == 4


def test_halo_volume_follows_the_cut(pkg):
    """mode="halo" moves the boundary only: a graph of P dense communities with a few cross edges
    needs a handful of rows per peer; the random graph needs (almost) every row of every shard."""
    import scipy.sparse as sp
    D = pkg.dist
    n, P = 64, 4
    rows = n // P
    comm = sp.block_diag([sp.random(rows, rows, density=0.5, format="csr", dtype=np.float32, random_state=k) for k in range(P)])
    cross = sp.csr_matrix((np.ones(3, np.float32), ([1, 20, 40], [17, 3, 63])), shape=(n, n))
    M = sp.csr_matrix(comm + cross + sp.eye(n, dtype=np.float32))
    A = pkg.csr_matrix(M.indptr.astype(np.uint32), M.indices.astype(np.uint32), M.data, n)
    V = D.halo_volume_matrix(A, P)
    assert V.sum() == 3 and V[0, 1] == 1 and V[1, 0] == 1 and V[2, 3] == 1 and (np.diag(V) == 0).all()
    ip, ix, dv = _graph(n, 5)
    V2 = D.halo_volume_matrix(pkg.csr_matrix(ip, ix, dv, n), P)
    off = V2[~np.eye(P, dtype=bool)]
    assert (off >= 0.5 * rows).all() and (off <= rows).all() and off.mean() > 4 * V.sum() / 12


# ------------------------------------------------------------------------------------------------
# rank-local load + partitioner hook
# ------------------------------------------------------------------------------------------------
def _community_graph(n, P, deg, p_in, seed):
    """P hidden communities (vertex ids shuffled), `deg` out-edges per vertex, a share p_in of them inside the
    vertex's own community: the kind of graph a partitioner pays on (the random synthetic Reddit does not)."""
    import scipy.sparse as sp
    rng = np.random.default_rng(seed)
    comm = rng.permutation(n) % P
    rows = np.repeat(np.arange(n), deg)
    same = rng.random(n * deg) < p_in
    cols = np.empty(n * deg, dtype=np.int64)
    for k in range(P):
        idx = np.nonzero((comm[rows] == k) & same)[0]
        cols[idx] = rng.choice(np.nonzero(comm == k)[0], size=idx.size)
    idx = np.nonzero(~same)[0]
    cols[idx] = rng.integers(0, n, size=idx.size)
    A = sp.csr_matrix((np.ones(n * deg, np.float32), (rows, cols)), shape=(n, n))
    A.sum_duplicates()
    A.data[:] = 1.0
    return A


def _rank_local_worker(rank, P, port, dirname, out_q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import __graft_entry__ as ge
    pkg = ge.load_package()
    D = pkg.dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=P)
    try:
        comm = D.host_comm()
        A_rows, AT_rows, X, Y, info = D.load_rank_local_host(comm, dirname)
        # what the whole-graph path builds for this rank (src/main.cpp:143-149)
        (ip, ix, dv, n, _), Xf, Yf, _ = pkg.datasets.read_dataset(dirname)
        assert np.all(dv == np.round(dv))                  # unit weights, as the reference's data-prep writes them
        A = pkg.csr_matrix(ip, ix, dv, n)
        A.normalize(True)
        A_T = A.transpose()
        p = D.partition_bounds(n, P)
        ok = True
        for got, full in ((A_rows, A), (AT_rows, A_T)):
            lo, hi = int(full.indptr[p[rank]]), int(full.indptr[p[rank + 1]])
            ok &= np.array_equal(got.indptr, full.indptr[p[rank]:p[rank + 1] + 1] - full.indptr[p[rank]])
            ok &= np.array_equal(got.indices, full.indices[lo:hi])
            # unit edge weights (what prep.py writes, test/data/prep.py:113): the column sums are small integers, exact
            # in any summation order -> the rank-local values are BITWISE the whole-graph ones (non-unit weights:
            # fp64 partial sums here vs fp32 row-order sums there, equal to an ulp -- documented in the loader)
            ok &= np.array_equal(got.data, full.data[lo:hi])
        ok &= np.array_equal(X, Xf[p[rank]:p[rank + 1]]) and np.array_equal(Y, Yf[p[rank]:p[rank + 1]])
        ok &= info["num_labels"] == 1 + int(Yf.max()) and info["n"] == n
        # the block matrices built from the row block == built from the whole matrix
        full_d = D.dist_row_csr_matrix(comm, A_T, p, p, 2)
        loc_d = D.dist_row_csr_matrix(comm, AT_rows, p, p, 2, row_block=True)
        for a, b in zip([full_d.diag, full_d.remote] + full_d.blocks + full_d.remote_chunks,
                        [loc_d.diag, loc_d.remote] + loc_d.blocks + loc_d.remote_chunks):
            ok &= np.array_equal(a.indptr, b.indptr) and np.array_equal(a.indices, b.indices) and a.m() == b.m()
        # halo send lists by exchange == the lists computed from the whole matrix
        need = D.halo_need_lists(loc_d.blocks, rank)
        got = comm.host_all_to_all([x.astype(np.int64).reshape(-1, 1) for x in need])
        for s in range(P):
            want = np.unique(D.split_row_block(A_T, p[s], p[s + 1], p)[rank].indices) if s != rank else np.empty(0)
            ok &= np.array_equal(got[s].reshape(-1), want.astype(np.int64))
        out_q.put((rank, bool(ok), sum(len(x) for x in need), info["host_bytes"]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("P", [2, 4])
def test_rank_local_load_and_partitioner_hook(tmp_path, P):
    """(a) prepare_dataset(partitioner="blocks") on a community graph: the halo volume of the partitioned files is
    below 50 % of what the all-gather moves ((P-1) n rows) -- and well below the unpartitioned graph's;
    (b) load_rank_local_host over gloo: every rank reads ONLY its rows of graph.bin / features.bin / labels.bin,
    normalises by an all-reduce of partial column sums and transposes by one all-to-all, and ends up with exactly
    the row blocks (and block matrices, and halo send lists) the whole-graph path of src/main.cpp:143-149 builds."""
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    pkg = ge.load_package()
    ds = pkg.datasets
    n, deg = 2048, 14
    A = _community_graph(n, P, deg, 0.95, seed=5)
    rng = np.random.default_rng(6)
    X = rng.standard_normal((n, 10)).astype(np.float32)
    Y = rng.integers(0, 7, size=n)
    plain = ds.prepare_dataset(str(tmp_path / "comm"), A, X, Y, P=P)
    parted = ds.prepare_dataset(str(tmp_path / "comm"), A, X, Y, P=P, partitioner="blocks")
    assert parted.endswith(os.path.join("partitioned", "comm"))
    vol = {}
    for name, d in (("plain", plain), ("parted", parted)):
        ip, ix, _, nn, _ = ds.read_csr(os.path.join(d, "graph.bin"))
        L = ds.comm_volume_matrix(ip, ix, P)
        vol[name] = int(L.sum() - np.trace(L))
    allgather = (P - 1) * n
    assert vol["parted"] < 0.5 * allgather and vol["parted"] < 0.6 * vol["plain"], (vol, allgather)
    # explicit permutation file (the reference's -p <file>, test/data/prep.py:241-243) gives the same files
    perm = ds.partition_blocks(ds.read_csr(os.path.join(plain, "graph.bin"))[:3], P)
    pf = tmp_path / "perm.txt"
    pf.write_text(" ".join(str(int(v)) for v in perm))
    again = ds.prepare_dataset(str(tmp_path / "again"), A, X, Y, P=P, permutation=str(pf))
    assert open(os.path.join(again, "graph.bin"), "rb").read() == open(os.path.join(parted, "graph.bin"), "rb").read()

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_local_worker, args=(r, P, port, parted, q)) for r in range(P)]
    for pr in procs:
        pr.start()
    res = sorted([q.get(timeout=180) for _ in range(P)])
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    assert all(ok for _, ok, _, _ in res), res
    import scipy.sparse as sp                                                  # the workers listed the forward matrix A^T
    ip, ix, dv, nn, _ = ds.read_csr(os.path.join(parted, "graph.bin"))
    T = sp.csr_matrix(sp.csr_matrix((dv, ix, ip), shape=(nn, nn)).T)
    LT = ds.comm_volume_matrix(T.indptr, T.indices, P)
    assert sum(rows for _, _, rows, _ in res) == int(LT.sum() - np.trace(LT))   # halo rows == the comm-volume matrix
    whole = os.path.getsize(os.path.join(parted, "graph.bin")) + os.path.getsize(os.path.join(parted, "features.bin"))
    assert max(b for _, _, _, b in res) < 0.9 * whole * 2                       # a rank holds its share, not P copies
