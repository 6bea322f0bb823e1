"""BASELINE.json configs C2..C5 at THEIR OWN sizes through the C ABI (-m gpu).

C2  Reddit shape, one GPU: one full training epoch (608-128-128-128-41) against the CPU oracle
    (loss, every G_W / G_b at 1e-4), plus fp64 sampled-row checks of the SpMMs whose plan decisions
    only trigger at this size (hot-column test, 6144/4096-row panels, 64/32 MiB slices, LPE pick,
    32 768-task rounds): d = 128 forward and backward matrix, d = 41 forward and backward matrix
    (the narrow form, plan built exactly as gcn_layer builds it).
C3  Reddit shape, 8 ranks: one rank's share (diagonal block + K = 4 pieces of the merged remote
    block, the all-gather schedule of dist.py), d = 128 and d = 48 (41 classes padded to x8,
    src/main.cpp:135).
C4  ogbn-products shape (n = 2 449 032, 126.2 M non-zeros): the whole SpMM and one rank's share.
C5  ogbn-papers100M shape: one rank's share at P = 8 (13.9 M x 111 M block, 216 M non-zeros)
    against the full all-gathered B (56.9 GB) resident in HBM.

The graphs are the synthetic stand-ins of SURVEY.md 8(d) (no dataset can exist on the box).
Row-partition results have no reference test: "parity unpinned by the
reference" -- they are pinned here against fp64 sums of the same products.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-4
SIZES = [608, 128, 128, 128, 41]


def relerr(got, want):
    want = np.asarray(want, dtype=np.float64)
    return float(np.abs(np.asarray(got, dtype=np.float64) - want).max() / (np.abs(want).max() + 1e-30))


@pytest.fixture(scope="module")
def ctx(pkg):
    return pkg.context(0)


def _with_matrices(pkg, data):
    """adds the two normalised matrices to a conftest stand-in (cached on the dict)"""
    if "A" not in data:
        A = pkg.csr_matrix(data["ip"], data["ix"], data["dv"].copy(), data["n"])
        A.normalize(True)                        # backward matrix  A D^-1
        data["A"], data["A_T"] = A, A.transpose()  # forward matrix  (A D^-1)^T
    return data


@pytest.fixture(scope="module")
def reddit(pkg):
    """the bench workload (SURVEY.md 8(d) stand-in): graph + features + labels, and the two normalised matrices"""
    from conftest import reddit_standin
    return _with_matrices(pkg, reddit_standin(pkg, "asym"))


@pytest.fixture(scope="module")
def reddit_both(pkg, reddit_any):
    """both stand-ins in turn: the SURVEY one and the symmetric one with the real dataset's structure"""
    return _with_matrices(pkg, reddit_any)


def sample_rows(M, k, rng, lo=0, hi=None):
    """k distinct rows of M[lo:hi] including the 8 heaviest (sliced) ones and an empty one if any"""
    hi = M.n() if hi is None else hi
    deg = np.diff(M.indptr[lo:hi + 1].astype(np.int64))
    rows = rng.choice(hi - lo, size=k, replace=False)
    rows[:8] = np.argsort(deg)[-8:]
    rows[8] = int(np.argmin(deg))
    return np.unique(rows) + lo


def rows_fp64(M, B, rows):
    """fp64 reference of (M B)[rows] from the host CSR (B: host array or a callable rows -> fp64 rows)"""
    out = np.zeros((len(rows), B.shape[1]), dtype=np.float64)
    for k, r in enumerate(rows):
        s, e = int(M.indptr[r]), int(M.indptr[r + 1])
        if e > s:
            out[k] = (M.data[s:e].astype(np.float64)[:, None] * B[M.indices[s:e].astype(np.int64)].astype(np.float64)).sum(axis=0)
    return out


def assert_rows_close(got, want, what):
    den = np.maximum(np.abs(want).max(axis=1), 1e-2 * np.abs(want).max()) + 1e-30
    err = (np.abs(got - want).max(axis=1) / den).max()
    assert err <= TOL, (what, float(err))


# ------------------------------------------------------------------------------------------------
# C2
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("fused", [True, False])
def test_c2_full_epoch_matches_oracle(pkg, oracle, ctx, reddit_both, fused):
    """src/gcn.hpp:437-489, :785-818 at BASELINE.json configs[1]: forward, loss, backward of the whole
    model; fused = the kernels bench.py times, unfused = the reference's launch sequence.  On BOTH stand-ins.
    The judge is the oracle twice: the fp32 restatement (what the CPU baseline times) and its exact-accumulation
    twin (oracle.Gcn(f64acc=True): same algorithm, every SpMM / GEMM sum in fp64, rounded once).  Over
    K = n = 232 968 terms the fp32 restatement's own sequential sums are 1.5e-4 away from the exact ones (layer 0
    G_b = 1^T G: measured) -- more than the 1e-4 bar -- so the device is judged against the exact twin and
    against the fp32 restatement within the restatement's own distance.  ~7 s + ~5 s per stand-in, cached."""
    from conftest import reddit_oracle_epoch
    reddit = reddit_both
    n = reddit["n"]
    exact = reddit_oracle_epoch(oracle, reddit, SIZES[-1], True)
    f32 = reddit_oracle_epoch(oracle, reddit, SIZES[-1], False)
    G = pkg.gcn(pkg.csr_matrix(reddit["ip"], reddit["ix"], reddit["dv"].copy(), n), SIZES, fused=fused)
    for layer, (W, b) in zip(G.layers(), exact["W"]):                     # same seed-99 init, bit for bit
        np.testing.assert_array_equal(layer.W().numpy(), W)
        np.testing.assert_array_equal(layer.b().numpy(), b)
    Xd, Yd = pkg.dn_matrix.from_numpy(reddit["X"]), pkg.dn_matrix.from_numpy(reddit["Y"])
    loss, acc = G.train_forward(ctx, Xd, Yd)
    G.backward(ctx)
    ctx.sync()
    for want in (exact, f32):
        assert abs(loss - want["loss"]) <= TOL * abs(want["loss"]), (loss, want["loss"])
        assert abs(acc - want["acc"]) <= 8.0 / n, (acc, want["acc"])      # near-ties may flip an argmax
    for li, layer in enumerate(G.layers()):
        for k, (what, got) in enumerate((("G_W", layer.GW().numpy()), ("G_b", layer.Gb().numpy()))):
            e = relerr(got, exact["grads"][li][k])
            assert e <= TOL, (reddit["kind"], li, what, e)
            own = relerr(f32["grads"][li][k], exact["grads"][li][k])      # the fp32 restatement's own error
            assert relerr(got, f32["grads"][li][k]) <= TOL + own, (reddit["kind"], li, what, own)
    # the plans the model built are the full-size forms: sweep tasks for both matrices
    for layer in G.layers():
        assert layer.A.ext_buffer.num_sweep_tasks() > 0
    del G


@pytest.mark.parametrize("which", ["forward", "backward"])
@pytest.mark.parametrize("d", [128, 41])
def test_c2_spmm_sampled_rows_fp64(pkg, ctx, reddit_both, which, d):
    """d = 128: spmm_sweep_pair_kernel (71 % of the epoch); d = 41: sweep_repack + spmm_sweep_quad_lds_kernel
    (15 %), both on the hot-column forward matrix and on the power-law-row backward matrix, with the plan
    the layer would build (get_matmul_buffer with the layer's width; max_d = 128 as sparse_linear asks)."""
    reddit = reddit_both
    n = reddit["n"]
    M = reddit["A_T"] if which == "forward" else reddit["A"]
    rng = np.random.default_rng(d + (which == "forward"))
    B = rng.standard_normal((n, d), dtype=np.float32)
    Bd, C = pkg.dn_matrix.from_numpy(B), pkg.dn_matrix(n, d)
    buf = pkg.get_matmul_buffer(ctx, M, Bd, C, max_d=128)
    assert buf.num_sweep_tasks() > 0
    ctx.fill(C, float("nan"))                                  # beta = 0 must not read C
    pkg.matmul(ctx, M, Bd, C, buf, 1.0, 0.0)
    ctx.sync()
    rows = sample_rows(M, 512, rng)
    got = C.numpy()
    assert np.isfinite(got).all()
    assert_rows_close(got[rows].astype(np.float64), rows_fp64(M, B, rows), (which, d))
    # every row: forward matrix is row-stochastic (A_fwd 1 = 1); backward matrix keeps column sums
    ones = pkg.dn_matrix(n, d); ctx.fill(ones, 1.0)
    if which == "forward":
        pkg.matmul(ctx, M, ones, C, buf, 1.0, 0.0); ctx.sync()
        assert float((C.t - 1.0).abs().max().item()) <= 2e-5
    else:
        colsum = got.astype(np.float64).sum(axis=0)
        assert np.abs(colsum - B.astype(np.float64).sum(axis=0)).max() <= 1e-4 * np.abs(B).sum(axis=0).max()
    # beta = 1 on top of the fused leaky-ReLU epilogue flag off/on a second time: C2 = M B + C (every slice chain)
    C0 = rng.standard_normal((n, d), dtype=np.float32)
    C2 = pkg.dn_matrix.from_numpy(C0)
    pkg.matmul(ctx, M, Bd, C2, buf, 1.0, 1.0); ctx.sync()
    assert_rows_close(C2.numpy()[rows].astype(np.float64), rows_fp64(M, B, rows) + C0[rows], (which, d, "beta"))


# ------------------------------------------------------------------------------------------------
# C3 / C4: one rank's share of the 1D row partition, all-gather schedule in K pieces
# ------------------------------------------------------------------------------------------------
def rank_share_check(pkg, ctx, M, P, r, K, d, seed, n_sample=256):
    """rows p[r]..p[r+1] of  M B  computed the way dist_sparse_linear (mode="allgather") computes them on
    rank r -- diagonal block on the own shard, then the K pieces of the merged remote block on the K
    gathered pieces (src/cuda_utils.hpp:57-92 regrouped) -- against fp64 rows of the global product."""
    D = pkg.dist
    n = M.n()
    p = D.partition_bounds(n, P)
    rows = p[r + 1] - p[r]
    diag, remote = D.split_local_remote(M, p[r], p[r + 1])
    assert diag.nnz() + remote.nnz() == int(M.indptr[p[r + 1]]) - int(M.indptr[p[r]])
    cb = D.chunk_bounds(rows, K)
    chunks = D.split_remote_chunks(remote, P, rows, K)
    rng = np.random.default_rng(seed)
    B = rng.standard_normal((n, d), dtype=np.float32)
    shard = pkg.dn_matrix.from_numpy(B[p[r]:p[r + 1]])
    # piece c of the exchange = rows cb[c]..cb[c+1] of EVERY shard, in rank order (one all-gather each)
    gathered = [pkg.dn_matrix.from_numpy(np.concatenate([B[p[s] + cb[c]:p[s] + cb[c + 1]] for s in range(P)]))
                for c in range(K)]
    C = pkg.dn_matrix(rows, d)
    ctx.fill(C, float("nan"))
    plan = lambda m: pkg.ops.spmm_plan_for(ctx, m, max(d, 128), d)        # dist_sparse_linear._plan
    pkg.ops._spmm(ctx, diag, shard, C, plan(diag), 1.0, 0.0)
    for c in range(K):
        pkg.ops._spmm(ctx, chunks[c], gathered[c], C, plan(chunks[c]), 1.0, 1.0)
    ctx.sync()
    got = C.numpy()
    assert np.isfinite(got).all()
    sel = sample_rows(M, n_sample, rng, p[r], p[r + 1])
    assert_rows_close(got[sel - p[r]].astype(np.float64), rows_fp64(M, B, sel), ("share", P, r, K, d))
    # all rows: the share of  M 1  is the vector of row sums
    ones_s = pkg.dn_matrix(rows, d); ctx.fill(ones_s, 1.0)
    pkg.ops._spmm(ctx, diag, ones_s, C, plan(diag), 1.0, 0.0)
    for c in range(K):
        og = pkg.dn_matrix(gathered[c].n(), d); ctx.fill(og, 1.0)
        pkg.ops._spmm(ctx, chunks[c], og, C, plan(chunks[c]), 1.0, 1.0)
    ctx.sync()
    rs = np.add.reduceat(np.concatenate([M.data[M.indptr[p[r]]:M.indptr[p[r + 1]]].astype(np.float64), [0.0]]),
                         (M.indptr[p[r]:p[r + 1]] - M.indptr[p[r]]).astype(np.int64))
    rs[np.diff(M.indptr[p[r]:p[r + 1] + 1].astype(np.int64)) == 0] = 0.0
    assert np.abs(C.numpy().astype(np.float64) - rs[:, None]).max() <= 3e-5 * max(1.0, np.abs(rs).max())
    return dict(diag_tasks=plan(diag).num_sweep_tasks(), piece_tasks=plan(chunks[0]).num_sweep_tasks())


@pytest.mark.parametrize("d", [128, 48])
def test_c3_reddit_rank_share_p8(pkg, ctx, reddit, d):
    """BASELINE.json configs[2]: Reddit across 8 ranks, 29 121 rows per rank; forward matrix, rank 3
    (left and right remote parts both present), K = 4 pieces (dist.default_chunks(8))."""
    K = pkg.dist.default_chunks(8)
    assert K == 4
    rank_share_check(pkg, ctx, reddit["A_T"], 8, 3, K, d, seed=30 + d)


def test_c3_reddit_rank_share_backward_matrix(pkg, ctx, reddit):
    rank_share_check(pkg, ctx, reddit["A"], 8, 0, 4, 128, seed=33)


def test_c4_products_shape(pkg, ctx):
    """BASELINE.json configs[3] (public OGB shape: n = 2 449 032 after padding, ~126.2 M non-zeros with both
    directions + self-loops, mean degree 51): B = 1.25 GB >> Infinity Cache; the sweep form is gated off
    (runs of one entry), the row-split form runs.  Whole-graph SpMM + one rank's share at P = 8."""
    n, nnz, d = 2_449_032, 126_200_000, 128
    ip, ix, dv = pkg.datasets.synth_powerlaw_csr(n, nnz, 17_500, seed=5)
    A = pkg.csr_matrix(ip, ix, dv, n)
    A.normalize(True)
    M = A.transpose()
    del A
    rng = np.random.default_rng(7)
    B = rng.standard_normal((n, d), dtype=np.float32)
    Bd, C = pkg.dn_matrix.from_numpy(B), pkg.dn_matrix(n, d)
    buf = pkg.get_matmul_buffer(ctx, M, Bd, C)
    assert buf.num_sweep_tasks() == 0           # mean run per panel << 2: the row-split form
    pkg.matmul(ctx, M, Bd, C, buf, 1.0, 0.0); ctx.sync()
    rows = sample_rows(M, 256, rng)
    assert_rows_close(C.numpy()[rows].astype(np.float64), rows_fp64(M, B, rows), "products")
    import torch
    B2, C2 = pkg.dn_matrix.from_numpy(2 * B), pkg.dn_matrix(n, d)
    pkg.matmul(ctx, M, B2, C2, buf, 1.0, 0.0); ctx.sync()
    assert torch.equal(C2.t, 2 * C.t)                                      # linearity, exact for a power of two
    del B2, C2, Bd, C, buf
    rank_share_check(pkg, ctx, M, 8, 5, 4, d, seed=41)


def test_c4_products_full_epoch_matches_oracle(pkg, oracle, ctx, tmp_path):
    """BASELINE.json configs[3] as a PATH, not a product: one full training epoch (forward, loss, backward) of the
    128-128-128-128-48 GCN on the products-shaped graph (undirected like the OGB one: pattern A = A^T, rows sorted;
    n = 2 449 032, 126.2 M non-zeros; 47 classes padded to 48 as at P = 8, src/main.cpp:135) on one GPU -- row-split
    SpMM (the sweep form is gated off at mean degree 51), GEMMs with M = 2.45 M rows, the fused loss over 2.45 M rows --
    against the exact-accumulation oracle: loss and every G_W / G_b at 1e-4."""
    (ip, ix, dv), X, Y = pkg.datasets.synth_products_like(1.0, seed=5, symmetric=True)
    n = ip.shape[0] - 1
    assert (n, int(ip[-1])) == (2_449_032, 126_200_000)
    sizes = [X.shape[1], 128, 128, 128, 48]
    O = oracle.Gcn(oracle.Csr(ip, ix, dv, n), sizes, f64acc=True)
    want_loss, want_acc = O.train_forward(X, Y)
    O.backward()
    want = [(l.lin.G_W.copy(), l.lin.G_b.copy()) for l in O.layers]
    del O
    G = pkg.gcn(pkg.csr_matrix(ip, ix, dv.copy(), n), sizes, fused=True)
    Xd, Yd = pkg.dn_matrix.from_numpy(X), pkg.dn_matrix.from_numpy(Y)
    loss, acc = G.train_forward(ctx, Xd, Yd)
    G.backward(ctx)
    ctx.sync()
    assert abs(loss - want_loss) <= TOL * abs(want_loss), (loss, want_loss)
    assert abs(acc - want_acc) <= 16.0 / n
    for li, (layer, (gw, gb)) in enumerate(zip(G.layers(), want)):
        assert relerr(layer.GW().numpy(), gw) <= TOL, (li, "G_W")
        assert relerr(layer.Gb().numpy(), gb) <= TOL, (li, "G_b")
        assert layer.A.ext_buffer.num_sweep_tasks() == 0            # the row-split form is what ran
    loss1, _ = G.train_step(ctx, Xd, Yd, 1e-2, 0.9, 0.999, 5e-4, 1e-8)
    loss2, _ = G.train_step(ctx, Xd, Yd, 1e-2, 0.9, 0.999, 5e-4, 1e-8)
    assert loss1 == pytest.approx(loss, rel=1e-6) and loss2 < loss1     # same forward again, then it trains
    del G, Xd, Yd
    import torch
    torch.cuda.empty_cache()
    # ... and as BASELINE.json states it: `mg_gcn -P 8 -R 1` (src/main.cpp:134-170) on the same files -- the partition,
    # the K-piece exchange, libmggcn_comm and the fused all-reduce at the products size, eight ranks wrapped over this
    # box's one GPU (peer-copy transport).  47 classes are padded to 48 there: the same function as the model above.
    import os, subprocess
    d = tmp_path / "permuted" / "products"
    pkg.datasets.write_dataset(str(d), ip, ix, dv, X, Y)
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mg-gcn_amd", "bin", "mg_gcn")
    r = subprocess.run([exe, "-P", "8", "-R", "1", "-E", "2", "train", str(d), "3", "128", "128", "128"], cwd=str(tmp_path),
                       env=dict(os.environ, MGGCN_OVERSUBSCRIBE="1"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stderr.strip().splitlines()
    assert lines[0] == "2449032 126200000" and lines[1] == "num_labels = 47" and lines[2] == "feature size = 128"
    ep = [ln.split() for ln in lines[3:5]]
    assert abs(float(ep[0][1]) - want_loss) <= TOL * abs(want_loss), (ep[0], want_loss)
    assert float(ep[1][1]) < float(ep[0][1])


def test_c5_papers100m_rank_share(pkg, ctx):
    """BASELINE.json configs[4]: rank 0's share at P = 8 -- [13 882 495 x 111 059 960], 216.25 M non-zeros
    (1.73 G / 8), power-law rows, uniformly random columns -- against the FULL gathered B [n x 128]
    (56.9 GB) resident in HBM.  66 GB of the 288 GB in use.  B is addressed with 64-bit offsets (row-split
    form).  256 sampled rows in fp64, every row through M 1 = row sums, linearity."""
    import torch
    P, n, d = 8, 111_059_960, 128
    rows, nnz = n // P, 1_730_000_000 // P
    ip, _, dv = pkg.datasets.synth_powerlaw_csr(rows, nnz, 20000, seed=3, self_loops=False)
    ix = np.random.default_rng(4).integers(0, n, size=nnz, dtype=np.uint32)
    dv = (dv / np.float32(15.6)).astype(np.float32)
    M = pkg.csr_matrix(ip, ix, dv, n)
    free0, _ = torch.cuda.mem_get_info()
    B = pkg.dn_matrix(n, d)
    with torch.cuda.stream(ctx.cuda_streams[0]):
        B.t.normal_()                                       # test input only
    C = pkg.dn_matrix(rows, d)
    buf = pkg.get_matmul_buffer(ctx, M, B, C)
    pkg.matmul(ctx, M, B, C, buf, 1.0, 0.0); ctx.sync()
    free1, _ = torch.cuda.mem_get_info()
    assert (free0 - free1) > 60e9                           # the 288 GB sizing claim: the whole B is resident
    rng = np.random.default_rng(5)
    sel = sample_rows(M, 256, rng)
    got = C.t[torch.from_numpy(sel).to(C.t.device)].cpu().numpy().astype(np.float64)
    want = np.zeros((len(sel), d))
    for k, r in enumerate(sel):
        s, e = int(ip[r]), int(ip[r + 1])
        if e > s:
            cols = torch.from_numpy(ix[s:e].astype(np.int64)).to(B.t.device)
            want[k] = (B.t[cols].double().cpu().numpy() * dv[s:e].astype(np.float64)[:, None]).sum(axis=0)
    assert_rows_close(got, want, "papers share")
    C1 = C.t.clone()
    with torch.cuda.stream(ctx.cuda_streams[0]):
        B.t.mul_(2.0)
    pkg.matmul(ctx, M, B, C, buf, 1.0, 0.0); ctx.sync()
    assert torch.equal(C.t, 2 * C1)
    del C1
    with torch.cuda.stream(ctx.cuda_streams[0]):
        B.t.fill_(1.0)
    pkg.matmul(ctx, M, B, C, buf, 1.0, 0.0); ctx.sync()
    rs = np.add.reduceat(np.concatenate([dv.astype(np.float64), [0.0]]), ip[:-1].astype(np.int64))
    rs[np.diff(ip.astype(np.int64)) == 0] = 0.0
    got1 = C.t[:, 0].double().cpu().numpy()
    assert np.abs(got1 - rs).max() <= 3e-5 * np.abs(rs).max()
    assert bool((C.t == C.t[:, :1]).all().item())          # every column of M 1 is the same vector


def _time_spmm(pkg, ctx, M, d, calls=5, reps=3, check_rows=0):
    Bh = np.random.default_rng(d).standard_normal((M.m(), d), dtype=np.float32)
    B = pkg.dn_matrix.from_numpy(Bh)
    C = pkg.dn_matrix(M.n(), d)
    buf = pkg.get_matmul_buffer(ctx, M, B, C, max_d=128)
    for _ in range(2):
        pkg.matmul(ctx, M, B, C, buf, 1.0, 0.0)
    ts = []
    for _ in range(reps):
        ctx.sync(); ctx.record("t_a", 0)
        for _ in range(calls):
            pkg.matmul(ctx, M, B, C, buf, 1.0, 0.0)
        ctx.record("t_b", 0); ctx.sync(); ctx.register_timer("t_ab", "t_a", "t_b")
        ts.append(ctx.measure("t_ab") / calls)
    if check_rows:                                  # ... and the result, on sampled rows in fp64
        rows = sample_rows(M, check_rows, np.random.default_rng(1))
        assert_rows_close(C.numpy()[rows].astype(np.float64), rows_fp64(M, Bh, rows), "order-invariance")
    return float(np.median(ts)), buf.describe()


def test_c2_spmm_time_does_not_depend_on_the_input_order(pkg, ctx):
    """Round 3's finding, kept from coming back: on the symmetric Reddit stand-in (rows sorted by column, power-law rows
    AND popular columns -- the real dataset's structure) the d = 128 SpMM took 3.09 ms when heavy rows were cut into
    contiguous slices, 2.34 with interleaved ones; the same rows in random order ran at 2.36 either way.  And a vertex
    numbering by decreasing degree (popular columns contiguous) is detected and handled by the plan's column
    permutation.  Generous 12 % margins: this guards the mechanism, not the last microsecond."""
    from conftest import reddit_standin
    data = reddit_standin(pkg, "sym")
    ip, ix, dv, n = data["ip"], data["ix"], data["dv"], data["n"]
    A = pkg.csr_matrix(ip, ix, dv.copy(), n)
    A.normalize(True)
    t_sorted, desc = _time_spmm(pkg, ctx, A, 128, check_rows=256)
    assert "permuted=0" in desc
    rng = np.random.default_rng(0)
    rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(ip.astype(np.int64)))
    order = np.argsort(rows * (1 << 32) + rng.integers(0, 1 << 32, size=ix.shape[0], dtype=np.int64), kind="stable")
    S = pkg.csr_matrix(ip, ix[order], A.data[order], n)              # same matrix, every row shuffled
    t_shuffled, _ = _time_spmm(pkg, ctx, S, 128)
    assert t_sorted <= 1.12 * t_shuffled, (t_sorted, t_shuffled)
    del S
    import scipy.sparse as sp
    deg = np.diff(ip.astype(np.int64))
    perm = np.argsort(-deg, kind="stable")
    M = sp.csr_matrix((A.data, ix, ip.astype(np.int64)), shape=(n, n))[perm][:, perm]
    M.sort_indices()
    D = pkg.csr_matrix(M.indptr.astype(np.uint32), M.indices.astype(np.uint32), M.data.astype(np.float32), n)
    t_degree, desc = _time_spmm(pkg, ctx, D, 128, check_rows=256)     # the column-permuted plan at full size, fp64 rows
    assert "permuted=1" in desc                                        # locality 0.11 >= 0.08
    assert t_degree <= 1.12 * t_shuffled, (t_degree, t_shuffled)
