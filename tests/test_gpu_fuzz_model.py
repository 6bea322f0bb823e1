"""Randomised model-level parity (-m gpu): the layer API of src/gcn.hpp on random graphs, layer stacks and options against the CPU
oracle -- loss, accuracy and every gradient of two epochs at 1e-4 (the replicas restart each epoch from the device's parameters,
as tests/test_gpu_gcn.py explains).  The fixed cases of test_gpu_gcn.py pin the shapes the reference runs; this walks the corners
nobody chose: one-layer models, width-1 hidden layers, two classes, graphs of eight vertices, a giant row, rows holding only their
self-loop, layer widths that grow and shrink (the SpMM-first / GEMM-first ordering rule of gcn.hpp:443-446 flips per layer)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-4


def relerr(got, want):
    return float(np.abs(got.astype(np.float64) - want.astype(np.float64)).max() / max(np.abs(want).max(), 1e-30))


def _random_graph(rng, n):
    """CSR with a self-loop on every row (test/data/prep.py:113 adds them; normalize(true) divides by the degree)"""
    law = rng.choice(["uniform", "power", "giant", "loops_mostly"])
    if law == "uniform":
        lens = rng.integers(0, min(n, 40), size=n)
    elif law == "power":
        lens = np.minimum((rng.pareto(1.2, size=n) * 4).astype(np.int64), n - 1)
    elif law == "giant":
        lens = rng.integers(0, 6, size=n)
        lens[int(rng.integers(0, n))] = n - 1
    else:
        lens = (rng.random(n) < 0.15) * rng.integers(1, 4, size=n)
    rows = []
    for r in range(n):
        others = rng.choice(n, size=int(min(lens[r], n)), replace=False) if lens[r] else np.zeros(0, np.int64)
        rows.append(np.unique(np.concatenate([others, [r]])))
    ip = np.concatenate([[0], np.cumsum([len(c) for c in rows])]).astype(np.uint32)
    ix = np.concatenate(rows).astype(np.uint32)
    return ip, ix, np.ones(len(ix), np.float32), law


def _sync(G, O):
    for layer, ol in zip(G.layers(), O.layers):
        ol.lin.W, ol.lin.b = layer.W().numpy().copy(), layer.b().numpy().copy()
        if layer.lin.mW is not None:
            ol.lin.mW, ol.lin.vW = layer.lin.mW.numpy().copy(), layer.lin.vW.numpy().copy()
            ol.lin.mb, ol.lin.vb = layer.lin.mb.numpy().copy(), layer.lin.vb.numpy().copy()
            ol.lin.step = layer.lin.step


@pytest.mark.parametrize("seed", range(36))
def test_random_model_matches_oracle(pkg, oracle, seed, monkeypatch):
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.choice([8, 16, 40, 64, 200, 520, 1500, 3000]))
    ip, ix, dv, law = _random_graph(rng, n)
    F = int(rng.choice([1, 2, 3, 16, 33, 100, 128, 608]))
    C = int(rng.choice([2, 3, 7, 41, 47]))
    hidden = [int(rng.choice([1, 2, 5, 16, 33, 64, 128, 200])) for _ in range(int(rng.integers(0, 4)))]
    sizes = [F] + hidden + [C]
    fused = bool(rng.integers(0, 2))
    if rng.random() < 0.4:                          # the sweep form and its column permutation on graphs this small
        monkeypatch.setenv("MGGCN_SPMM_SWEEP_MIN_NNZ", "1")
        monkeypatch.setenv("MGGCN_SPMM_SWEEP_MIN_RUN_X10", "0")
        monkeypatch.setenv("MGGCN_SPMM_PANEL_ROWS", str(int(rng.choice([64, 128]))))
        monkeypatch.setenv("MGGCN_SPMM_PANEL_ROWS_NARROW", "96")
        monkeypatch.setenv("MGGCN_SPMM_PERMUTE_COLUMNS", str(int(rng.integers(0, 2))))
    what = dict(seed=seed, n=n, law=law, sizes=sizes, fused=fused)
    X = rng.standard_normal((n, F), dtype=np.float32)
    Y = rng.integers(0, C, size=(n, 1)).astype(np.int32)
    ctx = pkg.context(0)
    G = pkg.gcn(pkg.csr_matrix(ip, ix, dv, n), sizes, fused=fused)
    O = oracle.Gcn(oracle.Csr(ip, ix, dv, n), sizes)
    Xd, Yd = pkg.dn_matrix.from_numpy(X), pkg.dn_matrix.from_numpy(Y)
    for epoch in range(2):
        _sync(G, O)
        loss, acc = G.train_forward(ctx, Xd, Yd)
        G.backward(ctx)
        ctx.sync()
        grads = [(l.GW().numpy().copy(), l.Gb().numpy().copy()) for l in G.layers()]
        G.adam_update(ctx, 1e-2, 0.9, 0.999, 5e-4, 1e-8)
        ctx.sync()
        ol, oa = O.train_forward(X, Y)
        O.backward()
        ograds = [(l.lin.G_W.copy(), l.lin.G_b.copy()) for l in O.layers]
        O.adam_update()
        assert np.isfinite(loss) and abs(loss - ol) <= TOL * abs(ol), (what, epoch, loss, ol)
        assert abs(acc - oa) <= 3.0 / n + 1e-9, (what, epoch, acc, oa)
        # gradients relative to the largest gradient of the model: a layer whose true gradient is rounding noise next to
        # the others (e.g. behind a width-1 bottleneck) has no 1e-4 of its own to be held to
        gmax = max(max(np.abs(w).max(), np.abs(b).max()) for w, b in ograds)
        for li in range(len(ograds)):
            for k, name in ((0, "G_W"), (1, "G_b")):
                got, want = grads[li][k], ograds[li][k]
                assert np.isfinite(got).all(), (what, epoch, li, name)
                err = np.abs(got.astype(np.float64) - want).max()
                assert err <= TOL * max(np.abs(want).max(), 1e-2 * gmax), (what, epoch, li, name, err, np.abs(want).max(), gmax)
