"""Parity of every HIP kernel behind the C ABI against the CPU oracle (the -m gpu
tests proper).  Floating-point bar: 1e-4 relative (BASELINE.json north_star), measured
as max|got - want| / max|want| per output and, for the SpMM, additionally row by row
against the fp64-accumulated oracle so that a wrong row cannot hide behind a large one.
Integer outputs (argmax, is_equal) are bit-exact."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-4


def relerr(got, want):
    want = np.asarray(want, dtype=np.float64)
    got = np.asarray(got, dtype=np.float64)
    return float(np.abs(got - want).max() / (np.abs(want).max() + 1e-30))


def rowwise_relerr(got, want):
    """max over rows of (row's max abs error) / (row's max abs value), the row scale floored at
    1 % of the global scale so that a row that cancels to ~0 (d = 1!) is judged against the
    magnitude of the numbers that were actually summed"""
    want = np.asarray(want, dtype=np.float64)
    got = np.asarray(got, dtype=np.float64)
    den = np.maximum(np.abs(want).max(axis=1), 1e-2 * np.abs(want).max()) + 1e-30
    return float((np.abs(got - want).max(axis=1) / den).max())


@pytest.fixture(scope="module")
def ctx(pkg):
    return pkg.context(0)


def _csr(pkg, oracle, ip, ix, dv, m):
    return pkg.csr_matrix(ip, ix, dv, m), oracle.Csr(ip, ix, dv, m)


def _run_spmm(pkg, ctx, A, B, C0, alpha, beta, flags=0, plan=True, max_d=None):
    Bd = pkg.dn_matrix.from_numpy(B)
    Cd = pkg.dn_matrix.from_numpy(C0)
    buf = pkg.get_matmul_buffer(ctx, A, Bd, Cd, alpha, beta, max_d=max_d) if plan else None
    pkg.matmul(ctx, A, Bd, Cd, buf, alpha, beta, flags)
    ctx.sync()
    return Cd.numpy(), buf


@pytest.mark.parametrize("d", [1, 4, 32, 41, 48, 64, 100, 128, 132, 256, 608])
@pytest.mark.parametrize("plan", [True, False])
def test_spmm_widths(pkg, oracle, ctx, d, plan):
    n = 1500
    ip, ix, dv = pkg.datasets.synth_powerlaw_csr(n, 60_000, 4000, seed=d)
    dv = np.random.default_rng(d).random(dv.shape[0], dtype=np.float32)
    A, Ao = _csr(pkg, oracle, ip, ix, dv, n)
    rng = np.random.default_rng(d + 1)
    B = rng.standard_normal((n, d), dtype=np.float32)
    C0 = rng.standard_normal((n, d), dtype=np.float32)
    for alpha, beta in [(1.0, 0.0), (0.5, 2.0), (1.0, 1.0)]:
        got, buf = _run_spmm(pkg, ctx, A, B, C0, alpha, beta, plan=plan)
        want = oracle.spmm(Ao, B, C0.copy(), alpha, beta, f64acc=True)
        assert rowwise_relerr(got, want) <= TOL, (d, alpha, beta)
        if plan:
            assert buf.num_split_rows() > 0       # max degree 4000 > split threshold: slices exercised


def test_spmm_baseline_config_c1(pkg, oracle, ctx):
    """BASELINE.json configs[0]: 10k nodes / 100k nnz, d = 128, column-normalised values."""
    n = 10_000
    ip, ix, dv = pkg.datasets.synth_uniform_csr(n, 10, seed=0)
    A, Ao = _csr(pkg, oracle, ip, ix, dv, n)
    A.normalize(True); oracle.normalize(Ao, True)
    np.testing.assert_array_equal(A.data, Ao.data)
    B = np.random.default_rng(0).standard_normal((n, 128), dtype=np.float32)
    got, _ = _run_spmm(pkg, ctx, A, B, np.zeros((n, 128), np.float32), 1.0, 0.0)
    assert rowwise_relerr(got, oracle.spmm(Ao, B, f64acc=True)) <= TOL
    assert relerr(got, oracle.spmm(Ao, B)) <= TOL                     # and vs the fp32 restatement


def test_spmm_edge_cases(pkg, oracle, ctx):
    # empty rows, a row longer than one 64-chunk, a row exactly 64 long, duplicates,
    # beta = 0 over NaN-filled C (never read), rectangular A
    rng = np.random.default_rng(5)
    lens = [0, 1, 63, 64, 65, 0, 128, 129, 700, 2, 0]
    ip = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint32)
    m = 37
    ix = rng.integers(0, m, size=int(ip[-1]), dtype=np.uint32)
    dv = rng.standard_normal(int(ip[-1])).astype(np.float32)
    A, Ao = _csr(pkg, oracle, ip, ix, dv, m)
    for d in (128, 41, 8):
        B = rng.standard_normal((m, d)).astype(np.float32)
        C0 = np.full((len(lens), d), np.nan, dtype=np.float32)
        for plan in (True, False):
            got, _ = _run_spmm(pkg, ctx, A, B, C0, 1.0, 0.0, plan=plan)
            want = oracle.spmm(Ao, B, f64acc=True)
            assert np.isfinite(got).all()
            assert np.abs(got - want).max() <= 1e-4 * np.abs(want).max()
            assert (got[[0, 5, 10]] == 0).all()
    # zero-row matrix: a no-op, not a crash
    E = pkg.csr_matrix([0], [], [], 5)
    pkg.matmul(ctx, E, pkg.dn_matrix.from_numpy(np.ones((5, 4), np.float32)), pkg.dn_matrix(0, 4), None, 1.0, 0.0)
    ctx.sync()


def test_spmm_fused_leaky_relu_and_linearity(pkg, oracle, ctx):
    n, d = 3000, 128
    ip, ix, dv = pkg.datasets.synth_powerlaw_csr(n, 150_000, 2500, seed=9)
    A, Ao = _csr(pkg, oracle, ip, ix, dv, n)
    A.normalize(True); oracle.normalize(Ao, True)
    rng = np.random.default_rng(10)
    B1 = rng.standard_normal((n, d), dtype=np.float32)
    B2 = rng.standard_normal((n, d), dtype=np.float32)
    z = np.zeros((n, d), np.float32)
    got, _ = _run_spmm(pkg, ctx, A, B1, z, 1.0, 0.0, flags=1)
    want = oracle.leaky_relu_forward(oracle.spmm(Ao, B1))
    assert relerr(got, want) <= TOL
    # size-independent property: A(B1 + 2 B2) == A B1 + 2 A B2
    s, _ = _run_spmm(pkg, ctx, A, B1 + 2 * B2, z, 1.0, 0.0)
    a, _ = _run_spmm(pkg, ctx, A, B1, z, 1.0, 0.0)
    b, _ = _run_spmm(pkg, ctx, A, B2, z, 1.0, 0.0)
    assert relerr(s, a + 2 * b) <= TOL
    # row-stochastic matrix times ones == ones  (checksum of the normalisation + kernel)
    At = A.transpose()
    o, _ = _run_spmm(pkg, ctx, At, np.ones((n, d), np.float32), z, 1.0, 0.0)
    np.testing.assert_allclose(o, 1.0, rtol=1e-5)


def test_spmm_is_bitwise_reproducible(pkg, ctx):
    n, d = 4000, 128
    ip, ix, dv = pkg.datasets.synth_powerlaw_csr(n, 300_000, 5000, seed=2)
    A = pkg.csr_matrix(ip, ix, dv, n)
    B = np.random.default_rng(3).standard_normal((n, d), dtype=np.float32)
    z = np.zeros((n, d), np.float32)
    a, _ = _run_spmm(pkg, ctx, A, B, z, 1.0, 0.0)
    b, _ = _run_spmm(pkg, ctx, A, B, z, 1.0, 0.0)
    np.testing.assert_array_equal(a, b)          # split rows are combined in a fixed order


GEMM_SHAPES = [  # (M, N, K, A_T, B_T)
    (300, 128, 608, False, False),      # H.W   (first layer)
    (300, 41, 128, False, False),       # logits layer, N not a multiple of 4
    (257, 128, 128, False, True),       # G.W^T
    (608, 128, 5000, True, False),      # G_W = X^T G, split-K
    (128, 41, 3001, True, False),       # split-K with ragged K and narrow N
    (1, 128, 4097, False, False),       # G_b = ones . G
    (700, 1, 41, False, False),         # softmax row sums (GEMM with a ones vector)
    (2, 2, 3, False, False), (2, 3, 2, True, False), (129, 65, 33, True, True),
]


@pytest.mark.parametrize("M,N,K,A_T,B_T", GEMM_SHAPES)
def test_gemm(pkg, oracle, ctx, M, N, K, A_T, B_T):
    rng = np.random.default_rng(M * 7 + N * 3 + K)
    A = rng.standard_normal((K, M) if A_T else (M, K)).astype(np.float32)
    B = rng.standard_normal((N, K) if B_T else (K, N)).astype(np.float32)
    C0 = rng.standard_normal((M, N)).astype(np.float32)
    for alpha, beta in [(1.0, 0.0), (1.0, 1.0), (-0.5, 0.25)]:
        Ad, Bd, Cd = (pkg.dn_matrix.from_numpy(x) for x in (A, B, C0))
        pkg.matmul(ctx, Ad, Bd, Cd, alpha, beta, A_T, B_T)
        ctx.sync()
        want = oracle.gemm(A, B, C0.copy(), alpha, beta, A_T, B_T, f64acc=True)
        assert relerr(Cd.numpy(), want) <= TOL, (alpha, beta)


def test_gemm_beta_zero_ignores_nan_and_shape_errors(pkg, ctx):
    A = pkg.dn_matrix.from_numpy(np.ones((5, 7), np.float32))
    B = pkg.dn_matrix.from_numpy(np.ones((7, 3), np.float32))
    C = pkg.dn_matrix.from_numpy(np.full((5, 3), np.nan, np.float32))
    pkg.matmul(ctx, A, B, C, 1.0, 0.0)
    ctx.sync()
    np.testing.assert_array_equal(C.numpy(), 7.0)
    with pytest.raises(ValueError):
        pkg.matmul(ctx, A, A, C, 1.0, 0.0)


def test_elementwise_kernels(pkg, oracle, ctx):
    ops = pkg.ops
    rng = np.random.default_rng(0)
    for shape in [(1000, 128), (333, 41), (7, 3)]:
        x = rng.standard_normal(shape).astype(np.float32)
        g = rng.standard_normal(shape).astype(np.float32)
        X, Gm = pkg.dn_matrix.from_numpy(x), pkg.dn_matrix.from_numpy(g)
        out = pkg.dn_matrix(shape)
        ops.leaky_relu_forward(ctx, X, out); ctx.sync()
        np.testing.assert_array_equal(out.numpy(), oracle.leaky_relu_forward(x))
        ops.leaky_relu_backward(ctx, X, Gm, out); ctx.sync()
        np.testing.assert_array_equal(out.numpy(), oracle.leaky_relu_backward(x, g))
        ops.leaky_relu_forward(ctx, X, X); ctx.sync()                      # in place (gcn.hpp:449)
        np.testing.assert_array_equal(X.numpy(), oracle.leaky_relu_forward(x))
        # broadcast_rows: discard and accumulate
        row = rng.standard_normal((1, shape[1])).astype(np.float32)
        R, M = pkg.dn_matrix.from_numpy(row), pkg.dn_matrix.from_numpy(x)
        ops.broadcast_rows(ctx, R, M, False); ctx.sync()
        np.testing.assert_array_equal(M.numpy(), x + row)
        ops.broadcast_rows(ctx, R, M, True); ctx.sync()
        np.testing.assert_array_equal(M.numpy(), np.broadcast_to(row, shape))
        # axpy / axpby / aaxpby / scale_mat
        Bm = pkg.dn_matrix.from_numpy(g)
        ops.axpby(ctx, pkg.dn_matrix.from_numpy(x), Bm, 0.1, 0.9); ctx.sync()
        np.testing.assert_allclose(Bm.numpy(), np.float32(0.1) * x + np.float32(0.9) * g, rtol=1e-6, atol=1e-7)
        Bm = pkg.dn_matrix.from_numpy(g)
        ops.aaxpby(ctx, pkg.dn_matrix.from_numpy(x), Bm, 0.001, 0.999); ctx.sync()
        np.testing.assert_allclose(Bm.numpy(), np.float32(0.001) * x * x + np.float32(0.999) * g, rtol=1e-6, atol=1e-7)
        Bm = pkg.dn_matrix.from_numpy(g)
        ops.axpy(ctx, pkg.dn_matrix.from_numpy(x), Bm, 5e-4); ctx.sync()
        np.testing.assert_allclose(Bm.numpy(), g + np.float32(5e-4) * x, rtol=1e-6, atol=1e-7)
        ops.scale_mat(ctx, Bm, 0.25); ctx.sync()
        np.testing.assert_allclose(Bm.numpy(), (g + np.float32(5e-4) * x) * np.float32(0.25), rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("n,m", [(5000, 41), (1000, 48), (300, 128), (64, 3), (10, 300)])
def test_loss_chain_and_fused_loss(pkg, oracle, ctx, n, m):
    import torch
    rng = np.random.default_rng(n + m)
    H = (rng.standard_normal((n, m)) * 3).astype(np.float32)
    H[0, :] = 1.5                                           # full tie: first index must win
    if m > 2:
        H[1, [1, 2]] = H[1].max() + 1                       # two-way tie
    Y = rng.integers(0, m, size=(n, 1)).astype(np.int32)
    ls, ac, G, O = oracle.softmax_cross_entropy(H, Y, n_global=2 * n)
    for fused in (False, True):
        L = pkg.softmax_cross_entropy_loss("t_", copy=True, fused=fused)
        Hd, Yd = pkg.dn_matrix.from_numpy(H), pkg.dn_matrix.from_numpy(Y)
        loss, acc = L(ctx, Hd, Yd, n_global=2 * n)
        assert abs(loss * n - ls) <= 1e-4 * abs(ls), fused
        assert round(acc * n) == round(ac), fused                     # integer-exact
        assert relerr(L.backward().numpy(), G) <= TOL, fused
        np.testing.assert_array_equal(Hd.numpy(), H)                   # copy=True leaves the logits alone
    # the unfused chain's intermediate kernels, bit-exact where integer
    Hd = pkg.dn_matrix.from_numpy(H)
    P = pkg.dn_matrix(n, 1, dtype=np.int32)
    pkg.ops.max_row_indices(ctx, Hd, P); ctx.sync()
    np.testing.assert_array_equal(P.numpy().reshape(-1), H.argmax(axis=1))
    mx = pkg.dn_matrix(n, 1)
    pkg.ops.max_rows(ctx, Hd, mx); ctx.sync()
    np.testing.assert_array_equal(mx.numpy().reshape(-1), H.max(axis=1))
    T = pkg.dn_matrix(n, 1)
    pkg.ops.is_equal(ctx, pkg.dn_matrix.from_numpy(Y), P, T); ctx.sync()
    np.testing.assert_array_equal(T.numpy().reshape(-1), (Y.reshape(-1) == H.argmax(axis=1)).astype(np.float32))
    s = torch.empty(1, dtype=torch.float32, device="cuda")
    pkg.ops.abssum(ctx, Hd, s); ctx.sync()
    assert abs(float(s.item()) - np.abs(H.astype(np.float64)).sum()) <= 1e-5 * np.abs(H).sum()


@pytest.mark.parametrize("m", [1, 2, 15, 16, 17, 32, 33, 41, 48, 49, 63, 64, 65, 128, 129])
def test_fused_loss_widths_out_of_place_and_reproducible(pkg, oracle, ctx, m):
    """every lanes-per-row / logits-per-lane form of the fused loss (16-lane rows up to 64 classes, a wave per row
    above), out of place == in place bit for bit, and the two reported scalars bitwise equal from run to run (they
    are per-workgroup partials summed in a fixed order, no atomics)"""
    import torch
    n = 4099                                                  # not a multiple of the rows in flight per wave
    rng = np.random.default_rng(7 * m + 1)
    H = (rng.standard_normal((n, m)) * 4).astype(np.float32)
    H[3, :] = -2.25                                           # full tie: first index wins
    Y = rng.integers(0, m, size=(n, 1)).astype(np.int32)
    ls, ac, G, _ = oracle.softmax_cross_entropy(H, Y, n_global=n)
    Yd = pkg.dn_matrix.from_numpy(Y)
    runs = []
    for _ in range(3):
        Hd, Gd = pkg.dn_matrix.from_numpy(H), pkg.dn_matrix(n, m)
        sums = torch.zeros(2, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()                              # torch zeroes on its own stream
        pkg.ops.softmax_xent_fused(ctx, Hd, Yd, 1.0 / n, sums, out=Gd); ctx.sync()
        np.testing.assert_array_equal(Hd.numpy(), H)          # the logits are left alone
        runs.append((Gd.numpy().copy(), sums.cpu().numpy().copy()))
    Hd = pkg.dn_matrix.from_numpy(H)
    sums = torch.zeros(2, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    pkg.ops.softmax_xent_fused(ctx, Hd, Yd, 1.0 / n, sums); ctx.sync()      # in place
    np.testing.assert_array_equal(Hd.numpy(), runs[0][0])
    for g, sm in runs[1:]:
        np.testing.assert_array_equal(g, runs[0][0])
        np.testing.assert_array_equal(sm, runs[0][1])          # bitwise: no atomics
    np.testing.assert_array_equal(sums.cpu().numpy(), runs[0][1])
    assert relerr(runs[0][0], G) <= TOL
    assert abs(float(runs[0][1][0]) - ls) <= 1e-4 * abs(ls)
    assert round(float(runs[0][1][1])) == round(ac)


def test_small_memset_kernel_and_host_scalars(pkg, ctx):
    """mggcn_memset_zero: word-aligned ranges go through the library's own kernel (device memory and mapped pinned host
    memory), everything else through the runtime; host_scalars are readable after a sync without a copy"""
    import torch
    for words in (1, 2, 255, 256, 257, 70000, 300000):
        t = torch.full((words + 2,), 7.0, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()                              # torch fills on ITS stream; the memset runs on the context's
        ctx.lib.mggcn_memset_zero(t.data_ptr() + 4, 4 * words, ctx.stream(0)); ctx.sync()
        h = t.cpu().numpy()
        assert h[0] == 7.0 and h[-1] == 7.0 and not h[1:-1].any(), words
    b = torch.full((16,), 255, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    ctx.lib.mggcn_memset_zero(b.data_ptr() + 1, 6, ctx.stream(0)); ctx.sync()          # unaligned: runtime path
    assert b.cpu().numpy().tolist() == [255] + [0] * 6 + [255] * 9
    hs = pkg.matrix.host_scalars(4)
    A = pkg.dn_matrix.from_numpy(np.array([[1.5, -2.5, 4.0]], dtype=np.float32))
    pkg.ops.abssum(ctx, A, hs[1:2]); ctx.sync()
    assert hs.numpy().tolist() == [0.0, 8.0, 0.0, 0.0]
    ctx.lib.mggcn_memset_zero(hs.data_ptr(), 16, ctx.stream(0)); ctx.sync()
    assert hs.numpy().tolist() == [0.0] * 4


def test_adam_fused_equals_chain_equals_oracle(pkg, oracle, ctx):
    for fused in (False, True):
        lin = pkg.linear("0_", 64, 32, True, fused)
        ol = oracle.Linear(64, 32)
        np.testing.assert_array_equal(lin.W.numpy(), ol.W)              # same seed-99 init
        np.testing.assert_array_equal(lin.b.numpy(), ol.b)
        rng = np.random.default_rng(1)
        for step in range(3):
            gw = rng.standard_normal((64, 32)).astype(np.float32)
            gb = rng.standard_normal((1, 32)).astype(np.float32)
            lin.G_W.init(gw); lin.G_b.init(gb)
            ol.G_W, ol.G_b = gw.copy(), gb.copy()
            lin.adam_update(ctx, 1e-2, 0.9, 0.999, 5e-4, 1e-8); ctx.sync()
            ol.adam_update(1e-2, 0.9, 0.999, 5e-4, 1e-8)
            assert relerr(lin.W.numpy(), ol.W) <= 1e-5
            assert relerr(lin.b.numpy(), ol.b) <= 1e-5


# ---- column-panel sweep form of the SpMM (spmm_sweep.hip) ---------------------------------
@pytest.fixture(params=["as-given", "permuted", "general"], ids=["columns-as-given", "columns-permuted", "general-pair-kernel"])
def force_sweep(monkeypatch, request):
    """Small test matrices would fall back to the row-split kernels: force the sweep form
    and a tiny panel so that the (panel, row) ordering of the entry streams is exercised.
    Second variant: the plan's internal column permutation (what a vertex order with locality triggers) forced on.
    Third: power-of-two pitches (d = 128, 256) through the general pair kernel instead of its FAST instance."""
    monkeypatch.setenv("MGGCN_SPMM_PERMUTE_COLUMNS", "1" if request.param == "permuted" else "0")
    if request.param == "general":
        monkeypatch.setenv("MGGCN_SPMM_FAST_PAIRS", "0")
    monkeypatch.setenv("MGGCN_SPMM_SWEEP_MIN_NNZ", "1")
    monkeypatch.setenv("MGGCN_SPMM_PANEL_ROWS", "64")
    monkeypatch.setenv("MGGCN_SPMM_PANEL_ROWS_NARROW", "96")
    monkeypatch.setenv("MGGCN_SPMM_SLICE_ROWS", "400")        # several column slices (beta chaining)
    monkeypatch.setenv("MGGCN_SPMM_SWEEP_MIN_RUN_X10", "0")   # sparse test rows would be sent to row-split
    monkeypatch.delenv("MGGCN_SPMM_ALGO", raising=False)


@pytest.mark.parametrize("d", [1, 8, 41, 48, 64, 66, 128, 130, 256, 608])
def test_sweep_spmm_widths(pkg, oracle, ctx, force_sweep, d):
    n = 1500
    ip, ix, dv = pkg.datasets.synth_powerlaw_csr(n, 60_000, 4000, seed=100 + d)
    dv = np.random.default_rng(d).random(dv.shape[0], dtype=np.float32)
    A, Ao = _csr(pkg, oracle, ip, ix, dv, n)
    rng = np.random.default_rng(d + 1)
    B = rng.standard_normal((n, d), dtype=np.float32)
    C0 = rng.standard_normal((n, d), dtype=np.float32)
    for alpha, beta, flags in [(1.0, 0.0, 0), (0.5, 2.0, 0), (1.0, 1.0, 1), (1.0, 0.0, 1)]:
        got, buf = _run_spmm(pkg, ctx, A, B, C0, alpha, beta, flags=flags)
        assert buf.num_sweep_tasks() > 0
        want = oracle.spmm(Ao, B, C0.copy(), alpha, beta, f64acc=True)
        if flags:
            want = oracle.leaky_relu_forward(want)
        assert rowwise_relerr(got, want) <= TOL, (d, alpha, beta, flags)


def test_sweep_spmm_edge_cases_and_reproducibility(pkg, oracle, ctx, force_sweep):
    rng = np.random.default_rng(5)
    lens = [0, 1, 7, 8, 9, 0, 128, 129, 700, 2, 0, 0, 5000, 3]     # empty rows, batch boundaries, a split row
    ip = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint32)
    m = 300
    ix = rng.integers(0, m, size=int(ip[-1]), dtype=np.uint32)
    dv = rng.standard_normal(int(ip[-1])).astype(np.float32)
    A, Ao = _csr(pkg, oracle, ip, ix, dv, m)
    for d in (128, 41):
        B = rng.standard_normal((m, d)).astype(np.float32)
        C0 = np.full((len(lens), d), np.nan, dtype=np.float32)
        got, buf = _run_spmm(pkg, ctx, A, B, C0, 1.0, 0.0)
        assert buf.num_sweep_tasks() > 0
        want = oracle.spmm(Ao, B, f64acc=True)
        assert np.isfinite(got).all()
        assert np.abs(got - want).max() <= 1e-4 * np.abs(want).max()
        assert (got[[0, 5, 10, 11]] == 0).all()
        again, _ = _run_spmm(pkg, ctx, A, B, C0, 1.0, 0.0)
        np.testing.assert_array_equal(got, again)                  # fixed fold order -> bitwise equal


@pytest.mark.parametrize("d_hint,d", [(41, 128), (41, 24), (41, 44), (128, 41), (16, 41), (64, 64), (3, 3)])
def test_sweep_plan_hint_only_picks_the_fast_form(pkg, oracle, ctx, force_sweep, d_hint, d):
    """mggcn_spmm_plan_create_for: a plan built for narrow rows (runs padded to four entries, B
    re-pitched through the plan's scratch when its pitch is not 16-byte) serves wide calls and
    the other way round; widths above the scratch fall back to the one-column-per-lane kernel."""
    n = 1200
    ip, ix, dv = pkg.datasets.synth_powerlaw_csr(n, 50_000, 3000, seed=7)
    dv = np.random.default_rng(7).standard_normal(dv.shape[0]).astype(np.float32)
    A, Ao = _csr(pkg, oracle, ip, ix, dv, n)
    rng = np.random.default_rng(d_hint * 1000 + d)
    B = rng.standard_normal((n, d), dtype=np.float32)
    C0 = rng.standard_normal((n, d), dtype=np.float32)
    ctx.set()
    h = ctx.lib.mggcn_spmm_plan_create_for(A.n(), A.m(), A.indptr.ctypes.data, A.indices.ctypes.data,
                                           A.data.ctypes.data, 128, d_hint)
    buf = pkg.ops.spmm_buffer(ctx.lib, h)
    assert buf.num_sweep_tasks() > 0
    for alpha, beta, flags in [(1.0, 0.0, 0), (0.5, 2.0, 1)]:
        Bd, Cd = pkg.dn_matrix.from_numpy(B), pkg.dn_matrix.from_numpy(C0)
        pkg.matmul(ctx, A, Bd, Cd, buf, alpha, beta, flags)
        ctx.sync()
        want = oracle.spmm(Ao, B, C0.copy(), alpha, beta, f64acc=True)
        if flags:
            want = oracle.leaky_relu_forward(want)
        assert rowwise_relerr(Cd.numpy(), want) <= TOL, (d_hint, d, alpha, beta)
        np.testing.assert_array_equal(Bd.numpy(), B)          # the re-pitch never writes the caller's B


def test_vertex_order_with_locality_gets_its_columns_permuted(pkg, oracle, ctx, monkeypatch):
    """A banded matrix (every row's columns within +-40 of the diagonal: an unpermuted community graph in miniature)
    is detected at plan time (`locality` in mggcn_spmm_plan_describe) and the plan relabels its columns; a matrix with
    scattered columns is left alone.  Same results either way."""
    monkeypatch.setenv("MGGCN_SPMM_SWEEP_MIN_NNZ", "1")
    monkeypatch.setenv("MGGCN_SPMM_PANEL_ROWS", "128")
    monkeypatch.setenv("MGGCN_SPMM_SWEEP_MIN_RUN_X10", "0")
    monkeypatch.delenv("MGGCN_SPMM_PERMUTE_COLUMNS", raising=False)
    monkeypatch.delenv("MGGCN_SPMM_ALGO", raising=False)
    n, deg = 6000, 24
    rng = np.random.default_rng(9)
    rows = np.repeat(np.arange(n), deg)
    banded = np.clip(rows + rng.integers(-40, 41, size=n * deg), 0, n - 1)
    scattered = rng.integers(0, n, size=n * deg)
    for cols, want_perm in ((banded, True), (scattered, False)):
        ip = (np.arange(n + 1) * deg).astype(np.uint32)
        ix = cols.astype(np.uint32)
        dv = rng.standard_normal(n * deg).astype(np.float32)
        A, Ao = _csr(pkg, oracle, ip, ix, dv, n)
        for d in (128, 41):
            B = rng.standard_normal((n, d), dtype=np.float32)
            C0 = rng.standard_normal((n, d), dtype=np.float32)
            got, buf = _run_spmm(pkg, ctx, A, B, C0, 0.5, 2.0)
            assert buf.num_sweep_tasks() > 0
            assert ("permuted=1" in buf.describe()) == want_perm, buf.describe()
            assert rowwise_relerr(got, oracle.spmm(Ao, B, C0.copy(), 0.5, 2.0, f64acc=True)) <= TOL


def test_sweep_equals_rowsplit_on_a_big_graph(pkg, ctx, monkeypatch):
    """> 2^20 non-zeros: the sweep form is picked by default; compare with the row-split form."""
    n, d = 40_000, 128
    ip, ix, dv = pkg.datasets.synth_powerlaw_csr(n, 3_000_000, 9000, seed=4)
    A = pkg.csr_matrix(ip, ix, dv, n)
    A.normalize(True)
    B = np.random.default_rng(3).standard_normal((n, d), dtype=np.float32)
    z = np.zeros((n, d), np.float32)
    monkeypatch.delenv("MGGCN_SPMM_ALGO", raising=False)
    s, bs = _run_spmm(pkg, ctx, A, B, z, 1.0, 0.0)
    assert bs.num_sweep_tasks() > 0                       # mean degree 75 over 40 k columns: dense enough
    monkeypatch.setenv("MGGCN_SPMM_ALGO", "rowsplit")
    r, br = _run_spmm(pkg, ctx, A, B, z, 1.0, 0.0)
    assert br.num_sweep_tasks() == 0
    assert rowwise_relerr(s, r) <= 2e-5


def test_plan_follows_in_place_edits_of_the_matrix(pkg, oracle, ctx, monkeypatch):
    """SpMM, normalize(True), SpMM with the SAME ext_buffer: a sweep plan carries its own copy of the
    values, the reference's cuSPARSE workspace does not (src/cuda_utils.hpp:94-102) -- the call sequence
    must multiply with the edited matrix both through a fresh get_matmul_buffer and through the old handle."""
    monkeypatch.delenv("MGGCN_SPMM_ALGO", raising=False)
    n, d = 40_000, 128
    ip, ix, dv = pkg.datasets.synth_powerlaw_csr(n, 8_000_000, 9000, seed=14)       # mean degree 100: dense enough for the sweep form
    dv = np.random.default_rng(14).random(dv.shape[0], dtype=np.float32) + 0.5
    A, Ao = _csr(pkg, oracle, ip, ix, dv.copy(), n)
    B = np.random.default_rng(15).standard_normal((n, d), dtype=np.float32)
    Bd, Cd = pkg.dn_matrix.from_numpy(B), pkg.dn_matrix(n, d)
    buf = pkg.get_matmul_buffer(ctx, A, Bd, Cd)
    assert buf.num_sweep_tasks() > 0
    pkg.matmul(ctx, A, Bd, Cd, buf, 1.0, 0.0); ctx.sync()
    assert rowwise_relerr(Cd.numpy(), oracle.spmm(Ao, B, f64acc=True)) <= TOL
    A.normalize(True); oracle.normalize(Ao, True)
    want = oracle.spmm(Ao, B, f64acc=True)
    pkg.matmul(ctx, A, Bd, Cd, buf, 1.0, 0.0); ctx.sync()                 # stale handle: re-planned inside
    assert rowwise_relerr(Cd.numpy(), want) <= TOL
    buf2 = pkg.get_matmul_buffer(ctx, A, Bd, Cd)
    assert buf2 is not buf and buf2.version == A._version
    pkg.matmul(ctx, A, Bd, Cd, buf2, 1.0, 0.0); ctx.sync()
    assert rowwise_relerr(Cd.numpy(), want) <= TOL
    A.data *= np.float32(2.0); A.invalidate()                             # any other in-place edit
    pkg.matmul(ctx, A, Bd, Cd, buf2, 1.0, 0.0); ctx.sync()
    assert rowwise_relerr(Cd.numpy(), 2 * want.astype(np.float64)) <= TOL


@pytest.mark.parametrize("M,N,K", [(1000, 128, 608), (777, 41, 128), (130, 48, 7), (5, 128, 9000)])
def test_gemm_bias_epilogue_equals_broadcast_then_gemm(pkg, oracle, ctx, M, N, K):
    """mggcn_gemm_bias_f32 = broadcast_rows + sgemm(beta = 1) of the reference's linear forward
    (src/gcn.hpp:116-123): same sum, bit for bit, in one pass (tile path and split-K path)."""
    rng = np.random.default_rng(M + N + K)
    X = rng.standard_normal((M, K)).astype(np.float32)
    W = rng.standard_normal((K, N)).astype(np.float32)
    b = rng.standard_normal((1, N)).astype(np.float32)
    Xd, Wd, bd = (pkg.dn_matrix.from_numpy(a) for a in (X, W, b))
    fused = pkg.dn_matrix.from_numpy(np.full((M, N), np.nan, dtype=np.float32))      # C is never read
    pkg.ops.linear_forward(ctx, Xd, Wd, bd, fused)
    two = pkg.dn_matrix(M, N)
    pkg.ops.broadcast_rows(ctx, bd, two, True)
    pkg.matmul(ctx, Xd, Wd, two, 1.0, 1.0)
    ctx.sync()
    np.testing.assert_array_equal(fused.numpy(), two.numpy())
    want = oracle.gemm(X, W, f64acc=True) + b
    assert rowwise_relerr(fused.numpy(), want) <= TOL


@pytest.mark.parametrize("d,lpe", [(41, 12), (41, 16), (48, 12), (24, 8), (24, 12), (24, 16), (8, 4), (8, 8), (8, 16), (3, 4), (64, 16)])
def test_sweep_narrow_forms_every_group_width(pkg, oracle, ctx, force_sweep, monkeypatch, d, lpe):
    """The narrow-row kernel fetches G = 64 / LPE rows per instruction (LPE = 4, 8, 12, 16 lanes of
    16 bytes per row); the plan normally picks LPE by padded stream length -- force each legal one."""
    monkeypatch.setenv("MGGCN_SPMM_NARROW_LPE", str(lpe))
    n = 1300
    ip, ix, dv = pkg.datasets.synth_powerlaw_csr(n, 45_000, 3000, seed=300 + d + lpe)
    dv = np.random.default_rng(d).standard_normal(dv.shape[0]).astype(np.float32)
    A, Ao = _csr(pkg, oracle, ip, ix, dv, n)
    rng = np.random.default_rng(lpe)
    B = rng.standard_normal((n, d), dtype=np.float32)
    C0 = rng.standard_normal((n, d), dtype=np.float32)
    for alpha, beta, flags in [(1.0, 0.0, 0), (0.5, 2.0, 1)]:
        got, buf = _run_spmm(pkg, ctx, A, B, C0, alpha, beta, flags=flags)
        assert buf.num_sweep_tasks() > 0
        want = oracle.spmm(Ao, B, C0.copy(), alpha, beta, f64acc=True)
        if flags:
            want = oracle.leaky_relu_forward(want)
        assert rowwise_relerr(got, want) <= TOL, (d, lpe, alpha, beta)
    again, _ = _run_spmm(pkg, ctx, A, B, C0, 0.5, 2.0, flags=1)
    np.testing.assert_array_equal(got, again)                      # reproducible


@pytest.mark.parametrize("d", [128, 41, 3])
def test_gather_rows_packs_the_halo(pkg, ctx, d):
    import torch
    rng = np.random.default_rng(d)
    src = rng.standard_normal((5000, d)).astype(np.float32)
    idx = rng.integers(0, 5000, size=1777).astype(np.int64)
    S = pkg.dn_matrix.from_numpy(src)
    D = pkg.dn_matrix.from_numpy(np.full((1777, d), np.nan, dtype=np.float32))
    dev_idx = torch.from_numpy(idx).to(torch.int32).to(ctx.device)
    pkg.ops.gather_rows(ctx, S, dev_idx, D)
    ctx.sync()
    np.testing.assert_array_equal(D.numpy(), src[idx])
    pkg.ops.gather_rows(ctx, S, dev_idx[:0], D)          # empty list: no-op


@pytest.mark.parametrize("M,N,K", [(1000, 128, 128), (2049, 128, 41), (300, 608, 128), (70, 48, 9000)])
def test_gemm_lrelu_backward_epilogue_equals_gemm_then_kernel(pkg, oracle, ctx, M, N, K):
    """mggcn_gemm_lrelu_bwd_f32 (G_out = (G . W^T) .* leaky_relu'(Z), src/gcn.hpp:135-137 + :462-468) is
    BITWISE the two launches it replaces (sgemm with beta = 0, then leaky_relu_backward_kernel), incl. the
    split-K path (K = 9000), and matches the oracle."""
    rng = np.random.default_rng(M + N + K)
    G = rng.standard_normal((M, K), dtype=np.float32)
    W = rng.standard_normal((N, K), dtype=np.float32)          # used transposed: G . W^T
    Z = rng.standard_normal((M, N), dtype=np.float32)
    Z[0, :4] = 0.0                                             # in > 0 is strict: zeros take the slope
    Gd, Wd, Zd = (pkg.dn_matrix.from_numpy(a) for a in (G, W, Z))
    fused = pkg.dn_matrix(M, N); ctx.fill(fused, float("nan"))
    pkg.ops.matmul_lrelu_backward(ctx, Gd, Wd, Zd, fused, 1.0, False, True)
    two = pkg.dn_matrix(M, N)
    pkg.matmul(ctx, Gd, Wd, two, 1.0, 0.0, False, True)
    pkg.ops.leaky_relu_backward(ctx, Zd, two, two)
    ctx.sync()
    np.testing.assert_array_equal(fused.numpy(), two.numpy())
    want = oracle.leaky_relu_backward(Z, oracle.gemm(G, W, B_T=True, f64acc=True))
    assert relerr(fused.numpy(), want) <= TOL
    np.testing.assert_array_equal(Zd.numpy(), Z)               # the mask operand is only read


def test_adam_multi_equals_per_tensor_fused(pkg, ctx):
    """mggcn_adam_multi_f32: one launch over a device table of all parameter tensors == mggcn_adam_fused_f32
    per tensor, bitwise, over several steps (sizes straddle the 1024-element blocks; biases take no decay)."""
    rng = np.random.default_rng(77)
    shapes = [(608, 128), (1, 128), (128, 128), (1, 128), (128, 41), (1, 41), (3, 1), (1025, 1)]
    mk = lambda a: pkg.dn_matrix.from_numpy(a.copy())
    P0 = [rng.standard_normal(s, dtype=np.float32) for s in shapes]
    sets = []
    for _ in range(2):
        sets.append([(mk(p), pkg.dn_matrix(*s), pkg.dn_matrix(*s), pkg.dn_matrix(*s)) for p, s in zip(P0, shapes)])
        for p, g, m, v in sets[-1]:
            m.zero(ctx); v.zero(ctx)
    wds = [5e-4 if s[0] > 1 else 0.0 for s in shapes]
    table = pkg.ops.adam_table(ctx, [(p, g, m, v, wd) for (p, g, m, v), wd in zip(sets[0], wds)])
    assert table.blocks == sum((s[0] * s[1] + 1023) // 1024 for s in shapes)
    for step in range(1, 4):
        bc1, bc2 = float(np.float32(1 - 0.9 ** step)), float(np.float32(1 - 0.999 ** step))
        grads = [rng.standard_normal(s, dtype=np.float32) for s in shapes]
        for k, gr in enumerate(grads):
            for st in sets:
                st[k][1].init(gr)
        table.step(ctx, 1e-2, 0.9, 0.999, bc1, bc2, 1e-8)
        for (p, g, m, v), wd in zip(sets[1], wds):
            pkg.ops.adam_fused(ctx, p, g, m, v, 1e-2, 0.9, 0.999, wd, bc1, bc2, 1e-8)
        ctx.sync()
        for a, b in zip(sets[0], sets[1]):
            for x, y in zip(a, b):
                np.testing.assert_array_equal(x.numpy(), y.numpy())


@pytest.mark.parametrize("n,din,dout", [(5000, 608, 128), (3001, 128, 41), (700, 16, 130), (9000, 128, 128), (40, 8, 5)])
def test_gemm_tn_colsum_equals_the_two_gemms(pkg, oracle, ctx, n, din, dout):
    """mggcn_gemm_tn_colsum_f32: G_W = X^T G and G_b = 1^T G in one pass (src/gcn.hpp:125-134) -- G_W BITWISE equal to
    the plain X^T G GEMM (same tiles, same split-K), G_b equal to the ones-vector GEMM / the oracle at 1e-4, incl.
    split-K shapes, two N-tiles (dout = 130) and ragged edges."""
    rng = np.random.default_rng(n + din)
    X = rng.standard_normal((n, din), dtype=np.float32)
    G = rng.standard_normal((n, dout), dtype=np.float32)
    Xd, Gd = pkg.dn_matrix.from_numpy(X), pkg.dn_matrix.from_numpy(G)
    GW, Gb = pkg.dn_matrix(din, dout), pkg.dn_matrix(1, dout)
    ctx.fill(GW, float("nan")); ctx.fill(Gb, float("nan"))
    pkg.ops.linear_backward_weights(ctx, Xd, Gd, GW, Gb)
    GW2, Gb2 = pkg.dn_matrix(din, dout), pkg.dn_matrix(1, dout)
    pkg.matmul(ctx, Xd, Gd, GW2, 1.0, 0.0, True)
    ones = pkg.dn_matrix(1, n); ctx.fill(ones, 1.0)
    pkg.matmul(ctx, ones, Gd, Gb2, 1.0, 0.0)
    ctx.sync()
    np.testing.assert_array_equal(GW.numpy(), GW2.numpy())
    want = oracle.gemm(np.ones((1, n), np.float32), G, f64acc=True)
    assert relerr(Gb.numpy(), want) <= TOL and relerr(Gb2.numpy(), want) <= TOL
    assert relerr(GW.numpy(), oracle.gemm(X, G, A_T=True, f64acc=True)) <= TOL


def test_spmm_random_shapes_forms_and_knobs(pkg):
    """A bounded run of profiles/experiments/spmm_fuzz_r03.py: random shapes (1 x 1 up to 9000 x 20000), degree laws (empty rows,
    giant rows, duplicates, one-column matrices), widths 1..608, alpha / beta / fused activation, NaN-filled C at beta = 0, and a
    random plan form per case (heuristics, forced sweep with random panel / slice / permutation / rows-per-task knobs, row-split,
    no plan) against scipy in fp64 at 1e-4 of the row's sum|a||b| budget.  (5000 cases of it ran clean in round 3:
    profiles/experiments/spmm_fuzz_r03.log, worst 4.1e-7.)"""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("spmm_fuzz", os.path.join(root, "profiles", "experiments", "spmm_fuzz_r03.py"))
    fuzz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fuzz)
    assert fuzz.main(cases=250, seed=2026) == 0


def test_gemm_random_shapes_transposes_and_epilogues(pkg):
    """A bounded run of profiles/experiments/gemm_fuzz_r03.py: mggcn_gemm_f32 (all transposes, alpha / beta, NaN-filled C at
    beta = 0), the bias, leaky-ReLU-backward and X^T G + column-sum epilogues over random (M, N, K) from 1 to 5000 x 5000 x 30000
    (odd sizes, split-K lengths, K below one MFMA step) against numpy in fp64 at 1e-4 of the entry's sum|a||b| budget.
    (1500 cases ran clean in round 3: profiles/experiments/gemm_fuzz_r03.log, worst 3.5e-7.)"""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("gemm_fuzz", os.path.join(root, "profiles", "experiments", "gemm_fuzz_r03.py"))
    fuzz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fuzz)
    assert fuzz.run(cases=200, seed=2026) == 0


def test_reserved_cus_shrink_the_launch_rounds_and_nothing_else(pkg, ctx, monkeypatch):
    """mggcn_spmm_plan_reserved_cus(16): plans built under it size their launch rounds to (CUs - 16) x 16 one-wave tasks (SpMMs that
    share the device with a collective kernel); the product is the same, and plans built afterwards are the
    plain ones again.  80 000 rows = more than one round of 16-row tasks either way."""
    import torch
    n, d = 80_000, 128
    ip, ix, dv = pkg.datasets.synth_powerlaw_csr(n, 8_000_000, 9000, seed=14)       # mean degree 100: dense enough for the sweep form
    A = pkg.csr_matrix(ip, ix, dv, n)
    A.normalize(True)
    B = np.random.default_rng(13).standard_normal((n, d), dtype=np.float32)
    z = np.zeros((n, d), np.float32)
    monkeypatch.delenv("MGGCN_SPMM_ALGO", raising=False)
    monkeypatch.delenv("MGGCN_SPMM_RESERVED_CUS", raising=False)
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    plain, bp = _run_spmm(pkg, ctx, A, B, z, 1.0, 0.0)
    assert bp.num_sweep_tasks() > 0 and bp.num_sweep_tasks() % (cus * 16) == 0
    ctx.lib.mggcn_spmm_plan_reserved_cus(16)
    try:
        A2 = pkg.csr_matrix(ip, ix, dv, n)                 # a fresh matrix: plans are cached per matrix
        A2.normalize(True)
        shared, bs = _run_spmm(pkg, ctx, A2, B, z, 1.0, 0.0)
    finally:
        ctx.lib.mggcn_spmm_plan_reserved_cus(0)
    assert bs.num_sweep_tasks() % ((cus - 16) * 16) == 0 and bs.num_sweep_tasks() != bp.num_sweep_tasks()
    assert rowwise_relerr(shared, plain) <= 2e-5
    A3 = pkg.csr_matrix(ip, ix, dv, n)
    A3.normalize(True)
    _, b3 = _run_spmm(pkg, ctx, A3, B, z, 1.0, 0.0)
    assert b3.num_sweep_tasks() == bp.num_sweep_tasks()
