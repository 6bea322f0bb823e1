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
