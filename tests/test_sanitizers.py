"""The engine's host-only code under AddressSanitizer + UndefinedBehaviorSanitizer and under ThreadSanitizer, on the
CPU: `make -C tests/native sanitize` builds tests/native/plan_host_test.cpp twice with plain g++ and
runs both.  Under test: the SpMM plan builders' host passes (mg-gcn_amd/csrc/plan_host.cpp -- threaded; the packed entry
streams are decoded and replayed as an SpMM against the oracle), the host preprocessing (mg-gcn_amd/csrc/host_prep.cpp
-- threaded; bit-exact against the oracle) and the per-GPU command queues of the single-process host layer
(mg-gcn_amd/host/enqueue.hpp).  Second binary, tests/native/comm_sim_test.cpp: the peer-copy transport of libmggcn_comm.so
(mg-gcn_amd/csrc/comm.cpp compiled as is) on a MODEL of HIP's stream / event semantics (tests/native/hipsim/): every rank a device
of its own, queued operations executed in random and adversarial orders, P enqueue threads under ThreadSanitizer, and mutation
runs that drop event waits (one at a time, one kind at a time) and must be noticed; the RCCL transport's multi-rank branches on a
model of RCCL's contract (a collective runs when every rank's matching call has reached the head of its stream).  Third, tests/native/abisim.cpp: the C ABI
itself as CPU loops on that model, under mg-gcn_amd/host/tests/test_dist.cpp compiled as is -- the C++ host layer's distributed
classes (enqueue threads, compute / communication streams and the event edges between them, the three exchange schedules,
dist_gcn training) with every rank on a device of its own, again under adversarial schedules, the sanitizers and wait-dropping
mutation runs.
GPU sanitizers do not exist on this pool; the device side is covered by the parity tests."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_code_is_clean_under_asan_ubsan_and_tsan():
    r = subprocess.run(["make", "-s", "-j4", "-C", os.path.join(ROOT, "tests", "native"), "sanitize"], capture_output=True, text=True,
                       timeout=1500)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-4000:]
    assert out.count("ALL PASSED") == 6, out[-4000:]                  # three binaries, once per sanitizer build
    assert out.count("TEST PASSED: host layer on the stream model") == 8 and out.count("waits the host layer's cases fail") == 5, out[-4000:]
    assert out.count("scenarios of the two transports on the stream model, 0 failed") == 2, out[-4000:]
    assert "NOT NOTICED" not in out and "DEADLOCK" not in out
    assert "TEST FAILED" not in out and "ERROR: AddressSanitizer" not in out and "WARNING: ThreadSanitizer" not in out
    assert "runtime error" not in out                                # UndefinedBehaviorSanitizer


@pytest.mark.parametrize("transport,mode,threads", [("p2p", "allgather", "1"), ("rccl", "rounds", "1"), ("rccl", "halo", "0")])
def test_cli_on_the_stream_model_matches_the_dist_oracle(pkg, oracle, tmp_path, transport, mode, threads):
    """`mg_gcn -P 4 -R 1 -E 2 train ...` -- host/main.cpp and the whole C++ host layer compiled as is over the CPU model of the C ABI
    (tests/native/abisim.cpp), four ranks on devices of their own, peer copies or the modelled RCCL, one enqueue thread per rank
    (ThreadSanitizer build) or the reference's single thread: the epoch-0 loss and accuracy equal oracle.DistGcn's at 1e-4 (classes
    padded to a multiple of P, src/main.cpp:135).  The host layer's LOGIC end to end without a GPU; the HIP kernels are the GPU
    suite's business."""
    exe = os.path.join(ROOT, "tests", "native", "_build", "mg_gcn_sim_tsan")
    r = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "native"), exe], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    n, F, C, P = 1536, 24, 6, 4
    ip, ix, dv = pkg.datasets.synth_powerlaw_csr(n, n * 20, 900, seed=17)
    rng = np.random.default_rng(18)
    X = rng.standard_normal((n, F), dtype=np.float32)
    Y = rng.integers(0, C, size=(n, 1)).astype(np.int32)
    Y[0, 0] = C - 1
    d = tmp_path / "permuted" / "synth"
    pkg.datasets.write_dataset(str(d), ip, ix, dv, X, Y)
    env = dict(os.environ, MGGCN_COMM_TRANSPORT=transport, MGGCN_DIST_MODE=mode, MGGCN_ENQUEUE_THREADS=threads, HIPSIM_POLICY="0",
               HIPSIM_SEED="9", TSAN_OPTIONS="halt_on_error=1")
    r = subprocess.run([exe, "-P", str(P), "-R", "1", "-E", "2", "train", str(d), "2", "16", "16"], cwd=str(tmp_path), env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ThreadSanitizer" not in r.stderr, r.stderr[-3000:]
    lines = r.stderr.strip().splitlines()
    got = [tuple(float(x) for x in ln.split()) for ln in lines if len(ln.split()) == 4 and ln.split()[0].isdigit()]
    assert [int(g[0]) for g in got] == [0, 1]
    O = oracle.DistGcn(oracle.Csr(ip, ix, dv, n), [F, 16, 16, C], P)
    want = O.train_forward(X, Y)
    assert abs(got[0][1] - want[0]) <= 1e-4 * want[0] and abs(got[0][2] - want[1]) <= 3.0 / n, (got, want)
    assert got[1][1] < got[0][1]                                      # and it trains
