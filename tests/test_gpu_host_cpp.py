"""The C++17 host layer (mg-gcn_amd/host: the reference's class names over the C ABI) on the
GPU: its restated reference tests and the `mg_gcn` CLI, whose per-epoch losses are checked
against the CPU oracle on a dataset written in the reference's on-disk format."""
import os
import re
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "mg-gcn_amd", "bin")


def _run(args, cwd=None, env=None):
    e = dict(os.environ)
    e.update(env or {})
    return subprocess.run(args, cwd=cwd, env=e, capture_output=True, text=True, timeout=600)


def test_cpp_reference_tests_pass(golden_dir):
    for exe, arg in (("test_matrix", golden_dir), ("test_gcn", os.path.join(golden_dir, "toyB"))):
        r = _run([os.path.join(BIN, exe), arg])
        assert r.returncode == 0, (exe, r.stdout, r.stderr)
        assert "TEST FAILED" not in r.stdout and r.stdout.count("TEST PASSED") >= 6, r.stdout


@pytest.mark.parametrize("fused", ["1", "0"])
def test_cli_matches_oracle(pkg, oracle, tmp_path, fused):
    n, F, C = 1024, 24, 6
    ip, ix, dv = pkg.datasets.synth_powerlaw_csr(n, n * 20, 900, seed=7)
    rng = np.random.default_rng(8)
    X = rng.standard_normal((n, F), dtype=np.float32)
    Y = rng.integers(0, C, size=(n, 1)).astype(np.int32)
    Y[0, 0] = C - 1
    d = tmp_path / "permuted" / "synth"
    pkg.datasets.write_dataset(str(d), ip, ix, dv, X, Y)
    r = _run([os.path.join(BIN, "mg_gcn"), "-E", "3", "train", str(d), "2", "16", "16"], cwd=str(tmp_path),
             env={"MGGCN_FUSED": fused, "MGGCN_DUMP_WEIGHTS": str(tmp_path / "w")})
    assert r.returncode == 0, r.stderr
    lines = r.stderr.strip().splitlines()
    assert lines[0] == f"{n} {n * 20}" and lines[1] == f"num_labels = {C}" and lines[2] == f"feature size = {F}"
    got = [tuple(float(x) for x in ln.split()) for ln in lines[3:6]]
    O = oracle.Gcn(oracle.Csr(ip, ix, dv, n), [F, 16, 16, C])
    want = []
    for _ in range(3):
        want.append(O.train_forward(X, Y)); O.backward(); O.adam_update()
    assert [int(g[0]) for g in got] == [0, 1, 2]
    assert abs(got[0][1] - want[0][0]) <= 1e-4 * want[0][0]            # identical inputs at epoch 0
    # every epoch at the 1e-4 bar from the exact state the CLI started it in (weights dumped by MGGCN_DUMP_WEIGHTS;
    # free-running trajectories are not comparable at 1e-4: Adam's lr * g / |g| turns the sign of a rounding-noise
    # gradient into a whole 0.01 step -- the optimiser's conditioning, not the kernels)
    O2 = oracle.Gcn(oracle.Csr(ip, ix, dv, n), [F, 16, 16, C])
    for e in range(3):
        for li, layer in enumerate(O2.layers):
            layer.lin.W = pkg.datasets.read_dense(str(tmp_path / "w" / f"e{e}_W{li}.bin"), "<f4").copy()
            layer.lin.b = pkg.datasets.read_dense(str(tmp_path / "w" / f"e{e}_b{li}.bin"), "<f4").copy()
        wl, wa = O2.train_forward(X, Y)
        assert abs(got[e][1] - wl) <= 1e-4 * wl and abs(got[e][2] - wa) <= 3.0 / n, (e, got[e], wl, wa)
    csv = tmp_path / "csvs" / f"permuted_synth_{F}_16_16_{C}_1.csv"       # reference file name scheme (main.cpp:100-111)
    text = csv.read_text()
    assert re.search(r"^0_0_0_0_matmul-spmm:", text, re.M) and re.search(r"^2_0_3_loss-layer:", text, re.M)


def test_cli_errors_like_the_reference(tmp_path):
    r = _run([os.path.join(BIN, "mg_gcn"), "bogus"], cwd=str(tmp_path))
    assert r.returncode == 1 and "uncaught exception: 'Unknown command.'" in r.stderr      # main.cpp:193, :198-206
    r = _run([os.path.join(BIN, "mg_gcn"), "train", str(tmp_path / "nope"), "1", "8"], cwd=str(tmp_path))
    assert r.returncode == 1 and "Aborting" in r.stderr
    r = _run([os.path.join(BIN, "mg_gcn"), "-h"], cwd=str(tmp_path))
    assert r.returncode == 0 and "Usage" in r.stdout


@pytest.mark.parametrize("P", [1, 2, 4])
def test_cpp_dist_classes_match_single_gpu(P):
    """host/tests/test_dist.cpp: dist_context / dist_row_csr_matrix / dist_row_dn_matrix / repl_dn_matrix /
    dist_gcn<true,...> + libmggcn_comm.so, every schedule (allgather in K pieces, halo, rounds) x overlap
    on/off (-S) x fused/unfused, against the single-GPU gcn.  P = 1 runs RCCL (one rank, bitwise equal to
    single GPU); P = 2, 4 wrap the ranks over this box's one GPU (MGGCN_OVERSUBSCRIBE=1: RCCL refuses two
    ranks per device, so the library's event-ordered peer-copy transport carries the exchange)."""
    r = _run([os.path.join(BIN, "test_dist"), str(P)], env={"MGGCN_OVERSUBSCRIBE": "1"} if P > 1 else {})
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    assert "TEST FAILED" not in r.stdout
    assert r.stdout.count("TEST PASSED") >= 12, r.stdout
    assert ("transport=rccl" in r.stdout) == (P == 1) and ("transport=p2p" in r.stdout) == (P > 1)


# (P, n, graph seed, F, C, hidden...): corners nobody chose -- shards of 8 rows, one hidden layer, none, width-1 layers, class
# counts that need padding to a multiple of P (src/main.cpp:135), eight ranks, K pieces longer than a shard has rows to give
DIST_SHAPES = [(2, 16, 1, 3, 2), (3, 393, 2, 1, 7, 5), (4, 64, 3, 33, 5, 1, 40), (8, 1024, 4, 16, 41, 128), (3, 1500, 5, 100, 3, 2, 2, 2),
               (8, 64, 6, 5, 9, 8), (6, 774, 7, 24, 47, 33, 17), (5, 1000, 8, 608, 4, 16), (2, 3000, 9, 7, 2, 200, 1, 64)]


@pytest.mark.parametrize("shape", DIST_SHAPES, ids=lambda s: "P%d-n%d-F%d-C%d-%s" % (s[0], s[1], s[3], s[4], "x".join(map(str, s[5:])) or "none"))
def test_cpp_dist_classes_on_odd_shapes(shape):
    """the same binary on shapes off the beaten path: every schedule x overlap x fused against the single-GPU model, two epochs at
    1e-4 (host/tests/test_dist.cpp takes `P n seed F C hidden...`)"""
    r = _run([os.path.join(BIN, "test_dist")] + [str(x) for x in shape], env={"MGGCN_OVERSUBSCRIBE": "1"})
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    assert "TEST FAILED" not in r.stdout
    assert r.stdout.count("TEST PASSED") >= 12, r.stdout


def test_cpp_dist_per_peer_pulling_streams():
    """MGGCN_P2P_PEER_STREAMS=1: every (receiver, sender) pair pulls on a stream of its own even when the ranks share a device --
    the form the peer-copy transport takes by itself only between DIFFERENT GPUs (one stream per xGMI link); forced here so that
    the fork / join event plumbing of csrc/comm.cpp runs on a one-GPU box: every schedule at 1e-4, bit-identical both ways."""
    r = _run([os.path.join(BIN, "test_dist"), "4"], env={"MGGCN_OVERSUBSCRIBE": "1", "MGGCN_P2P_PEER_STREAMS": "1"})
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    assert "TEST FAILED" not in r.stdout and r.stdout.count("TEST PASSED") >= 12, r.stdout


@pytest.mark.parametrize("P,extra", [(2, {}), (4, {}), (4, {"MGGCN_P2P_PEER_STREAMS": "1"}), (8, {})])
def test_cpp_dist_senders_push(P, extra):
    """MGGCN_P2P_PUSH=1: the peer-copy transport with the copies turned round -- the SENDER writes its piece into every
    receiver's buffer once that buffer is free, receivers only wait (csrc/comm.cpp: p2p_push / p2p_receive; a sender's release is
    local).  Same suite: every schedule at 1e-4 against the single-GPU model, enqueue threads and one thread bit-identical, the
    late rank still gets the original data; with one pushing stream per peer as between different GPUs, and without."""
    r = _run([os.path.join(BIN, "test_dist"), str(P)], env=dict({"MGGCN_OVERSUBSCRIBE": "1", "MGGCN_P2P_PUSH": "1"}, **extra))
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    assert "TEST FAILED" not in r.stdout and r.stdout.count("TEST PASSED") >= 12, r.stdout
    assert "transport=p2p-push" in r.stdout


def test_cpp_dist_senders_push_wraps_the_event_ring():
    """the push form with 40 pieces per SpMM: its own event slots (ready / pushed per pair) are re-used every 32 exchanges"""
    r = _run([os.path.join(BIN, "test_dist"), "4", "1536", "42", "24", "6", "32", "16"],
             env={"MGGCN_OVERSUBSCRIBE": "1", "MGGCN_DIST_CHUNKS": "40", "MGGCN_P2P_PUSH": "1"})
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    assert "TEST FAILED" not in r.stdout and r.stdout.count("TEST PASSED") >= 12, r.stdout


def test_cpp_dist_many_pieces_wrap_the_event_ring():
    """40 pieces per SpMM and ONE release at its end: more exchanges in a row than the peer-copy transport has event slots per
    rank (32) -- slots are re-used, ranks that are 16 exchanges behind with their releases release on the spot
    (csrc/comm.cpp); every schedule still matches the single-GPU model at 1e-4, enqueue threads and one thread bit-identical."""
    r = _run([os.path.join(BIN, "test_dist"), "4", "1536", "42", "24", "6", "32", "16"],
             env={"MGGCN_OVERSUBSCRIBE": "1", "MGGCN_DIST_CHUNKS": "40"})
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    assert "TEST FAILED" not in r.stdout and r.stdout.count("TEST PASSED") >= 12, r.stdout


@pytest.mark.parametrize("P,flags,mode", [(1, [], "allgather"), (2, [], "allgather"), (2, ["-S", "x"], "halo"),
                                          (4, [], "rounds")])
def test_cli_row_partition_matches_dist_oracle(pkg, oracle, tmp_path, P, flags, mode):
    """`mg_gcn -P <P> -R 1 [-S x] train ...` (src/main.cpp:134-170) through the distributed classes -- at
    -P 1 too -- against oracle.DistGcn (classes padded to a multiple of P, src/main.cpp:135)."""
    n, F, C = 1536, 24, 6
    ip, ix, dv = pkg.datasets.synth_powerlaw_csr(n, n * 20, 900, seed=17)
    rng = np.random.default_rng(18)
    X = rng.standard_normal((n, F), dtype=np.float32)
    Y = rng.integers(0, C, size=(n, 1)).astype(np.int32)
    Y[0, 0] = C - 1
    d = tmp_path / "permuted" / "synth"
    pkg.datasets.write_dataset(str(d), ip, ix, dv, X, Y)
    env = {"MGGCN_DIST_MODE": mode}
    if P > 1:
        env["MGGCN_OVERSUBSCRIBE"] = "1"
    r = _run([os.path.join(BIN, "mg_gcn"), "-P", str(P), "-R", "1", "-E", "2"] + flags + ["train", str(d), "2", "16", "16"],
             cwd=str(tmp_path), env=env)
    assert r.returncode == 0, r.stderr
    lines = r.stderr.strip().splitlines()
    got = [tuple(float(x) for x in ln.split()) for ln in lines[3:5]]
    O = oracle.DistGcn(oracle.Csr(ip, ix, dv, n), [F, 16, 16, C], P)
    want = O.train_forward(X, Y)
    assert abs(got[0][1] - want[0]) <= 1e-4 * want[0] and abs(got[0][2] - want[1]) <= 3.0 / n
    # the file name carries the UNPADDED class count: it is built (src/main.cpp:100-111) before the padding (:135)
    text = (tmp_path / "csvs" / f"permuted_synth_{F}_16_16_{C}_{P}.csv").read_text()
    assert re.search(rf"^0_{P - 1}_0_0_matmul-spmm:", text, re.M)          # per-rank timers "<epoch>_<rank>_<name>"


def test_cli_hoisted_first_aggregation_matches_the_plain_run(pkg, oracle, tmp_path):
    """MGGCN_HOIST_FIRST_AGGREGATION=1 (optional 6-SpMM epoch of the C++ host layer, same option as
    gcn(hoist_first_aggregation=True) in Python): epoch-0 loss equal to the oracle's and to the plain run's at 1e-4,
    and it trains."""
    n, F, C = 1024, 24, 6
    ip, ix, dv = pkg.datasets.synth_powerlaw_csr(n, n * 20, 900, seed=27)
    rng = np.random.default_rng(28)
    X = rng.standard_normal((n, F), dtype=np.float32)
    Y = rng.integers(0, C, size=(n, 1)).astype(np.int32)
    Y[0, 0] = C - 1
    d = tmp_path / "permuted" / "synth"
    pkg.datasets.write_dataset(str(d), ip, ix, dv, X, Y)
    O = oracle.Gcn(oracle.Csr(ip, ix, dv, n), [F, 16, 16, C])
    want = O.train_forward(X, Y)
    got = {}
    for hoist in ("0", "1"):
        r = _run([os.path.join(BIN, "mg_gcn"), "-E", "3", "train", str(d), "2", "16", "16"], cwd=str(tmp_path),
                 env={"MGGCN_HOIST_FIRST_AGGREGATION": hoist})
        assert r.returncode == 0, r.stderr
        got[hoist] = [tuple(float(x) for x in ln.split()) for ln in r.stderr.strip().splitlines()[3:6]]
        assert abs(got[hoist][0][1] - want[0]) <= 1e-4 * want[0]
        assert got[hoist][2][1] < got[hoist][0][1]
    assert abs(got["1"][0][1] - got["0"][0][1]) <= 1e-4 * got["0"][0][1]


def test_prep_files_train_through_the_cli(pkg, oracle, tmp_path):
    """edge list -> mg-gcn_amd/prep.py -> `mg_gcn -P 2 -R 1 train <dir>`: the files the prep front writes (padded, self-loops,
    permuted) are what the CLI reads -- header line, class count and feature width as written, epoch-0 loss as the oracle
    computes it from the SAME files (test/data/prep.py + src/main.cpp:82-98 end to end)."""
    import subprocess
    import sys
    rng = np.random.default_rng(5)
    n, E, F, C = 600, 9000, 21, 5
    e = rng.integers(0, n, size=(2, E))
    np.save(tmp_path / "e.npy", e)
    np.save(tmp_path / "x.npy", rng.standard_normal((n, F)).astype(np.float32))
    y = rng.integers(0, C, size=n); y[0] = C - 1
    np.save(tmp_path / "y.npy", y)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "mg-gcn_amd", "prep.py"), "--edges", str(tmp_path / "e.npy"),
                        "--features", str(tmp_path / "x.npy"), "--labels", str(tmp_path / "y.npy"),
                        "--out", str(tmp_path / "data" / "g"), "-P", "8", "--seed", "2"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    d = r.stdout.splitlines()[0]
    (ip, ix, dv, nn, _), X, Y, _ = pkg.datasets.read_dataset(d)
    assert nn == 600 and X.shape[1] == 24                                   # 600 = 8 * 75 already; 21 features -> 24
    r = _run([os.path.join(BIN, "mg_gcn"), "-P", "2", "-R", "1", "-E", "2", "train", d, "2", "16", "16"], cwd=str(tmp_path),
             env={"MGGCN_OVERSUBSCRIBE": "1"})
    assert r.returncode == 0, r.stderr
    lines = r.stderr.strip().splitlines()
    assert lines[0] == f"{nn} {len(ix)}" and lines[1] == f"num_labels = {C}" and lines[2] == "feature size = 24"
    got = [tuple(float(x) for x in ln.split()) for ln in lines[3:5]]
    O = oracle.DistGcn(oracle.Csr(ip, ix, dv, nn), [24, 16, 16, C], 2)      # classes padded to a multiple of P inside
    want = O.train_forward(X, Y)
    assert abs(got[0][1] - want[0]) <= 1e-4 * want[0] and got[1][1] < got[0][1]
