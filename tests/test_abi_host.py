"""CPU-side checks of the product: the C-ABI library loads and exports every symbol
include/mggcn.h declares (no device calls), and the host-side preprocessing
entry points (normalize / transpose / block split / weight init -- the
reference does this work on the host as well) agree with the oracle bit for bit."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "mggcn.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mggcn_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(pkg):
    names = _declared()
    assert len(names) >= 50
    lib = ctypes.CDLL(pkg._lib.LIB_PATH)
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    # the Python binding types exactly the declared set
    assert sorted(pkg._lib.PROTOTYPES) == names


def test_abi_version_and_no_gpu_is_loud(pkg):
    lib = pkg._lib.load()
    assert lib.mggcn_abi_version() == 1
    if lib.mggcn_device_count() == 0:
        with pytest.raises(pkg.engine_error):
            pkg.context(0)              # the product has no CPU path


def test_missing_library_is_loud(pkg, tmp_path):
    code = ("import sys; sys.path.insert(0, %r); import __graft_entry__ as g; p = g.load_package();"
            "p._lib.LIB_PATH = %r; p._lib._lib = None\n"
            "try:\n    p._lib.load()\nexcept p.engine_error as e:\n    print('LOUD', e); sys.exit(7)\n"
            % (ROOT, str(tmp_path / "nope.so")))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert r.returncode == 7 and "no CPU fallback" in r.stdout


def _rand_csr(n, m, dens, seed):
    import scipy.sparse as sp
    M = sp.random(n, m, density=dens, format="csr", dtype=np.float32, random_state=seed)
    M.data = M.data + 0.1
    return M


@pytest.mark.parametrize("axis", [False, True])
def test_normalize_host_matches_oracle(pkg, oracle, axis):
    M = _rand_csr(300, 200, 0.05, 1)
    a = pkg.csr_matrix(M.indptr, M.indices, M.data.copy(), 200)
    a.normalize(axis)
    b = oracle.Csr(M.indptr, M.indices, M.data.copy(), 200)
    oracle.normalize(b, axis)
    np.testing.assert_array_equal(a.data, b.data)


def test_normalize_host_threaded_is_close(pkg, oracle, monkeypatch):
    # > 1M non-zeros switches the threaded path on; column sums are combined in thread
    # order, so only the rounding of the partial sums may differ from the serial loop
    rng = np.random.default_rng(0)
    n, deg = 20000, 128
    ip = (np.arange(n + 1, dtype=np.uint64) * deg).astype(np.uint32)
    ix = rng.integers(0, n, size=n * deg, dtype=np.uint32)
    dv = rng.random(n * deg, dtype=np.float32) + 0.5
    monkeypatch.setenv("MGGCN_HOST_THREADS", "4")
    a = pkg.csr_matrix(ip, ix, dv.copy(), n); a.normalize(True)
    b = oracle.Csr(ip, ix, dv.copy(), n); oracle.normalize(b, True)
    np.testing.assert_allclose(a.data, b.data, rtol=2e-6)
    t = a.transpose()
    bt = oracle.transpose(oracle.Csr(ip, ix, a.data.copy(), n))
    np.testing.assert_array_equal(t.indptr, bt.indptr)
    np.testing.assert_array_equal(t.indices, bt.indices)      # serial (row-ascending) order
    np.testing.assert_array_equal(t.data, bt.data)


def test_transpose_host_matches_oracle(pkg, oracle):
    M = _rand_csr(257, 123, 0.1, 2)
    t = pkg.csr_matrix(M.indptr, M.indices, M.data, 123).transpose()
    o = oracle.transpose(oracle.Csr(M.indptr, M.indices, M.data, 123))
    assert (t.n(), t.m()) == (123, 257)
    np.testing.assert_array_equal(t.indptr, o.indptr)
    np.testing.assert_array_equal(t.indices, o.indices)
    np.testing.assert_array_equal(t.data, o.data)
    np.testing.assert_array_equal(t.as_dn(), M.toarray().T)


def test_transpose_empty_rows_and_cols(pkg):
    a = pkg.csr_matrix([0, 0, 2, 2], [0, 3], [1.0, 2.0], 5)
    t = a.transpose()
    assert t.indptr.tolist() == [0, 1, 1, 1, 2, 2] and t.indices.tolist() == [1, 1]


def test_block_split_host_matches_oracle(pkg, oracle):
    dist = pytest.importorskip("mg_gcn_amd.dist")
    M = _rand_csr(64, 64, 0.2, 3)
    p = [0, 16, 32, 48, 64]
    A = pkg.csr_matrix(M.indptr, M.indices, M.data, 64)
    want = oracle.block_split(oracle.Csr(M.indptr, M.indices, M.data, 64), p, p)
    for i in range(4):
        got = dist.split_row_block(A, p[i], p[i + 1], p)
        for j in range(4):
            np.testing.assert_array_equal(got[j].indptr, want[i][j].indptr)
            np.testing.assert_array_equal(got[j].indices, want[i][j].indices)
            np.testing.assert_array_equal(got[j].data, want[i][j].data)


def test_init_uniform_host_is_bit_identical_to_oracle(pkg, oracle):
    lib = pkg._lib.load()
    for (n, m, gain) in [(608, 128, -1.0), (128, 41, -1.0), (1, 128, float(np.sqrt(np.float32(1.0) / 3)))]:
        a = np.empty((n, m), dtype=np.float32)
        lib.mggcn_init_uniform_host(a.ctypes.data, n, m, gain)
        b = oracle.init_uniform(n, m, None if gain < 0 else gain)
        np.testing.assert_array_equal(a, b)


def test_format_roundtrip_and_errors(pkg, tmp_path):
    ds = pkg.datasets
    ip, ix, dv = ds.synth_uniform_csr(100, 5, seed=1)
    X = np.random.default_rng(0).standard_normal((100, 8)).astype(np.float32)
    Y = np.arange(100) % 7
    ds.write_dataset(str(tmp_path / "d"), ip, ix, dv, X, Y)
    (g, X2, Y2, S2) = ds.read_dataset(str(tmp_path / "d"))
    np.testing.assert_array_equal(g[0], ip); np.testing.assert_array_equal(g[1], ix)
    np.testing.assert_array_equal(g[2], dv); assert g[3:] == (100, 100)
    np.testing.assert_array_equal(X2, X); np.testing.assert_array_equal(Y2.reshape(-1), Y)
    assert Y2.dtype == np.int32 and S2.sum() == 0
    with pytest.raises(ds.format_error):
        ds.read_csr(str(tmp_path / "graph.txt"))            # "File type is not supported."
    bad = tmp_path / "bad.bin"; bad.write_bytes(b"PIGO-CSR-v1" + b"\x04\x04" + b"\0" * 16)
    with pytest.raises(ds.format_error):
        ds.read_csr(str(bad))
    trunc = tmp_path / "trunc.bin"
    trunc.write_bytes(open(tmp_path / "d" / "graph.bin", "rb").read()[:-5])
    with pytest.raises(ds.format_error):
        ds.read_csr(str(trunc))
    with pytest.raises(pkg.matrix_error):
        pkg.csr_matrix(str(tmp_path / "graph.txt"))


def test_synthetic_generators_hit_their_shapes(pkg):
    ds = pkg.datasets
    ip, ix, dv = ds.synth_uniform_csr(10_000, 10, seed=0)           # BASELINE.json configs[0]
    assert ip[-1] == 100_000 and ix.shape == (100_000,) and ix.max() < 10_000
    rows = np.repeat(np.arange(10_000), 10)
    assert len(set(zip(rows.tolist(), ix.tolist()))) == 100_000      # distinct columns per row
    ip, ix, dv = ds.synth_powerlaw_csr(4096, 200_000, 3000, seed=1)
    deg = np.diff(ip.astype(np.int64))
    assert ip[-1] == 200_000 and deg.min() >= 1 and deg.max() <= 3000 and deg.max() > 10 * deg.mean()
    assert np.array_equal(ix[ip[:-1]], np.arange(4096))              # self-loops
    (g, X, Y) = ds.synth_reddit_like(scale=0.002, seed=1)
    assert g[0].shape[0] - 1 == X.shape[0] == Y.shape[0] and X.shape[1] == 608 and Y.max() == 40
    assert X.shape[0] % 8 == 0


def test_prepare_dataset_pads_loops_and_permutes(pkg, tmp_path):
    """the reference's data-prep steps (test/data/prep.py:78-126) without DGL"""
    import scipy.sparse as sp
    ds = pkg.datasets
    n0, F0 = 13, 5
    rng = np.random.default_rng(0)
    adj = sp.random(n0, n0, density=0.3, format="csr", dtype=np.float32, random_state=1)
    adj.data[:] = 1.0
    X = rng.standard_normal((n0, F0)).astype(np.float32)
    Y = rng.integers(0, 4, size=n0)
    S = rng.integers(0, 3, size=n0)
    d = ds.prepare_dataset(str(tmp_path / "g"), adj, X, Y, S, P=8, seed=0)
    (ip, ix, dv, n, m), X2, Y2, S2 = ds.read_dataset(d)
    assert n == m == 16 and X2.shape == (16, 8)                         # padded to multiples of 8
    dense = sp.csr_matrix((dv, ix, ip), shape=(n, n)).toarray()
    assert (np.diag(dense) == 1).all()                                   # self-loops, padding vertices too
    off = dense - np.diag(np.diag(dense))
    want = adj.toarray() - np.diag(np.diag(adj.toarray()))
    np.testing.assert_array_equal(off[:n0, :n0], want)
    assert off[n0:].sum() == 0 and off[:, n0:].sum() == 0
    np.testing.assert_array_equal(X2[:n0, :F0], X); assert X2[n0:].sum() == 0 and X2[:, F0:].sum() == 0
    np.testing.assert_array_equal(Y2.reshape(-1)[:n0], Y); np.testing.assert_array_equal(S2.reshape(-1)[:n0], S)
    # permuted variant: same multiset of degrees / labels, written under permuted/
    d2 = ds.prepare_dataset(str(tmp_path / "g"), adj, X, Y, S, P=8, seed=3)
    assert os.path.normpath(d2).endswith(os.path.join("permuted", "g"))
    (ip2, ix2, dv2, _, _), X3, Y3, _ = ds.read_dataset(d2)
    assert sorted(np.diff(ip2).tolist()) == sorted(np.diff(ip).tolist())
    assert sorted(Y3.reshape(-1).tolist()) == sorted(Y2.reshape(-1).tolist())
    dense2 = sp.csr_matrix((dv2, ix2, ip2), shape=(n, n)).toarray()
    perm = np.random.default_rng(3).permutation(n)
    np.testing.assert_array_equal(dense2, dense[perm][:, perm])
    np.testing.assert_array_equal(X3, X2[perm])


def test_reserved_accumulator_registers_are_left_alone_by_the_compiler(tmp_path):
    """The float4 sweep kernels keep their accumulator planes in v[64:127] behind the register
    allocator's back (amdgpu_num_vgpr(64) + asm clobbers, csrc/spmm_sweep.hip).  Guard the
    contract at build time: in those kernels no compiler-generated instruction may name a VGPR
    >= 64, the kernels must be allotted 128 VGPRs, and nothing may spill to scratch."""
    import shutil
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    src = os.path.join(ROOT, "mg-gcn_amd", "csrc", "spmm_sweep.hip")
    out = tmp_path / "sweep.s"
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                        "-S", "--cuda-device-only", src, "-o", str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    text = out.read_text()
    hi = re.compile(r"\bv(6[4-9]|[7-9][0-9]|[12][0-9][0-9])\b|v\[(6[4-9]|[7-9][0-9]|[12][0-9][0-9]):")
    kern, inside, seen = None, False, set()
    for line in text.split("\n"):
        m = re.match(r"^(_Z\S*):", line)
        if m:
            kern = m.group(1)
        if "ASMSTART" in line:
            inside = True
            continue
        if "ASMEND" in line:
            inside = False
            continue
        if kern and ("spmm_sweep_pair_kernel" in kern or "spmm_sweep_quad_lds_kernel" in kern):
            seen.add("pair" if "pair" in kern else "quad")
            st = line.strip()
            if not inside and st and not st.startswith((".", ";")) and hi.search(st):
                raise AssertionError(f"{kern}: compiler-generated instruction touches a reserved register: {st}")
            # the fold's s_set_gpr_idx_on writes M0 behind the compiler's back: it must never hold a value there
            if not inside and st and not st.startswith((".", ";")) and re.search(r"\bm0\b", st.split(";")[0]):
                raise AssertionError(f"{kern}: compiler-generated instruction uses m0: {st}")
    assert seen == {"pair", "quad"}
    # the wave-priority rotation (rotate_priority: all four s_setprio levels) is compiled into both kernel families
    for fam in ("spmm_sweep_pair_kernel", "spmm_sweep_quad_lds_kernel"):
        bodies = re.findall(r"^(_Z\S*%s[^\s:]*):[^\n]*\n(.*?)s_endpgm" % fam, text, flags=re.S | re.M)
        assert bodies, fam
        for name, body in bodies:
            assert {int(x) for x in re.findall(r"s_setprio (\d)", body)} == {0, 1, 2, 3}, name
    # metadata of EVERY instantiation (pair<general|FAST> + quad_lds<4|8|12|16>): 128 VGPRs, no AGPRs (a spill of the
    # reserved planes would go there first), no scratch
    meta = text[text.index("amdhsa.kernels"):]
    blocks = [b for b in re.split(r"\n\s*- \.agpr_count:", "\n" + meta)[1:]]
    checked = 0
    for b in blocks:
        blk = ".agpr_count:" + b
        nm = re.search(r"\.name:\s+(\S+)", blk)
        if not nm or not ("spmm_sweep_pair_kernel" in nm.group(1) or "spmm_sweep_quad_lds_kernel" in nm.group(1)):
            continue
        checked += 1
        assert re.search(r"\.agpr_count:\s+0\b", blk), (nm.group(1), blk[:400])
        assert re.search(r"\.vgpr_count:\s+128\b", blk), (nm.group(1), blk[:400])
        assert re.search(r"\.private_segment_fixed_size:\s+0\b", blk), (nm.group(1), blk[:400])
    assert checked == 6, checked


def _longest_store_run(body: str, store_re: str) -> int:
    """longest run of store instructions with no s_waitcnt ... vmcnt between two of them"""
    best = cur = 0
    for line in body.split("\n"):
        st = line.split(";")[0].strip()
        if re.match(store_re, st):
            cur += 1
            best = max(best, cur)
        elif re.match(r"s_waitcnt\b.*vmcnt", st):
            cur = 0
    return best


def test_epilogues_issue_their_stores_back_to_back(tmp_path):
    """vmcnt counts stores: an epilogue whose stores sit inside per-element / per-row branches gets s_waitcnt vmcnt(0)
    in front of each of them from the compiler -- every store a dependent memory round trip (r02: 45 % of a GEMM wave's
    life at K = 128, sixteen serial round trips at the end of every SpMM wave).  Guard the straight-line forms: somewhere
    in each hot kernel a whole block's stores are issued with no vmcnt wait between them."""
    import shutil
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    want = {
        "spmm_sweep.hip": [(r"spmm_sweep_pair_kernel", r"global_store_dwordx4\b", 12),
                           (r"spmm_sweep_quad_lds_kernelILi16E", r"global_store_dword\b", 48)],
        "gemm.hip": [(r"gemm_mfma_kernelILb1ELb0ELi128ELi512ELb1E", r"global_store_dword\b", 16),
                     (r"gemm_mfma_kernelILb1ELb1ELi128ELi512ELb1E", r"global_store_dword\b", 16),
                     (r"gemm_mfma_kernelILb0ELb0ELi128ELi512ELb1E", r"global_store_dword\b", 16)],
    }
    for fname, kernels in want.items():
        out = tmp_path / (fname + ".s")
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                            "-S", "--cuda-device-only", os.path.join(ROOT, "mg-gcn_amd", "csrc", fname), "-o", str(out)],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        text = out.read_text()
        for pat, store_re, least in kernels:
            bodies = re.findall(r"^(_Z\S*%s[^\s:]*):[^\n]*\n(.*?)s_endpgm" % pat, text, flags=re.S | re.M)
            assert bodies, pat
            for name, body in bodies:
                run = _longest_store_run(body, store_re)
                assert run >= least, f"{name}: longest run of stores without a vmcnt wait is {run} (< {least})"


def test_comm_library_exports_every_declared_symbol():
    """include/mggcn_comm.h (the single-process multi-GPU exchange the C++ host layer links): every
    declared entry point is exported by libmggcn_comm.so.  Symbols are read from the ELF dynamic
    table -- loading the library would pull in RCCL and touch the GPU runtime."""
    text = open(os.path.join(ROOT, "include", "mggcn_comm.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = sorted(set(re.findall(r"\b(mggcn_comm_[a-z0-9_]+)\s*\(", text)))
    assert len(names) == 16, names          # 9 all-rank entry points + 4 per-rank ones + flags / release / release_rank
    lib = os.path.join(ROOT, "mg-gcn_amd", "lib", "libmggcn_comm.so")
    assert os.path.exists(lib), "build() must produce libmggcn_comm.so"
    nm = None
    for tool in ("nm", "/opt/rocm/lib/llvm/bin/llvm-nm"):
        r = subprocess.run([tool, "-D", "--defined-only", lib], capture_output=True, text=True)
        if r.returncode == 0:
            nm = r.stdout
            break
    assert nm is not None, "no nm tool"
    exported = set(re.findall(r"\b(mggcn_comm_[a-z0-9_]+)\b", nm))
    assert not [n for n in names if n not in exported], (names, sorted(exported))


def test_alltoallv_displacements_against_a_simulated_exchange(tmp_path):
    """csrc/comm_layout.h (the send / receive offsets both transports of mggcn_comm_alltoallv_f32 use; ADVICE r02: the
    RCCL branch has never run on more than one GPU): compiled on the host and checked against a numpy simulation of
    the exchange -- every rank packs its outgoing pieces in destination order, every receiver must find the piece
    of source k exactly where rdis says, in source order, with nothing overlapping and nothing left over."""
    src = tmp_path / "disp.cpp"
    src.write_text(
        '#include <cstdio>\n#include <cstdlib>\n#include <vector>\n#include "comm_layout.h"\n'
        'int main(int argc, char **argv) { int P = std::atoi(argv[1]); std::vector<std::size_t> c(P * P), s(P * P), r(P * P);\n'
        '  for (int i = 0; i < P * P; i++) c[i] = std::strtoull(argv[2 + i], nullptr, 10);\n'
        '  mggcn_layout::alltoallv_displacements(P, c.data(), s.data(), r.data());\n'
        '  for (auto x : s) std::printf("%zu ", x); std::printf("\\n"); for (auto x : r) std::printf("%zu ", x); std::printf("\\n"); }\n')
    exe = tmp_path / "disp"
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "mg-gcn_amd", "csrc"), str(src), "-o", str(exe)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    rng = np.random.default_rng(0)
    for P in (1, 2, 3, 8):
        counts = rng.integers(0, 7, size=(P, P))
        counts[rng.integers(0, P), :] = 0                      # a rank that sends nothing
        out = subprocess.run([str(exe), str(P)] + [str(int(x)) for x in counts.reshape(-1)], capture_output=True, text=True)
        assert out.returncode == 0
        sdis, rdis = (np.array(ln.split(), dtype=np.int64).reshape(P, P) for ln in out.stdout.strip().splitlines())
        # simulate: send[j] = pieces for k = 0..P-1 back to back, each element tagged (source, destination, index)
        send = [[(j, k, e) for k in range(P) for e in range(counts[j, k])] for j in range(P)]
        recv = [[None] * int(counts[:, k].sum()) for k in range(P)]
        for j in range(P):
            for k in range(P):
                piece = send[j][sdis[j, k]:sdis[j, k] + counts[j, k]]
                assert all(t[:2] == (j, k) for t in piece)
                for e, t in enumerate(piece):
                    assert recv[k][rdis[k, j] + e] is None     # no overlap
                    recv[k][rdis[k, j] + e] = t
        for k in range(P):
            assert all(t is not None for t in recv[k])         # nothing left over
            assert [t[0] for t in recv[k]] == sorted(t[0] for t in recv[k])      # source order


def test_public_headers_are_plain_c(tmp_path):
    """The drop-in boundary is a C ABI: include/*.h must compile as C99 (no HIP / torch / C++ types)."""
    src = tmp_path / "hdr.c"
    src.write_text('#include "mggcn.h"\n#include "mggcn_comm.h"\nint main(void) { return 0; }\n')
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only",
                        "-I", os.path.join(ROOT, "include"), str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_traffic_counters_belong_to_the_kernels_in_the_tree():
    """bench.py replays `roofline.traffic` from profiles/spmm_hbm_traffic.json (separate rocprofv3 --pmc passes of an
    earlier run).  The file carries a content hash of csrc/spmm*.hip at collection time: after a kernel change the
    counters have to be collected again (profiles/collect.sh + summarize.py) -- this test is what notices."""
    import json
    sys.path.insert(0, ROOT)
    import bench
    j = json.load(open(os.path.join(ROOT, "profiles", "spmm_hbm_traffic.json")))
    assert j.get("kernel_source_sha") == bench.spmm_kernel_sha(), \
        "profiles/spmm_hbm_traffic.json is older than csrc/spmm*.hip: run profiles/collect.sh + summarize.py again"
    assert os.path.exists(os.path.join(ROOT, j["source"]))


def test_edge_list_to_dataset_files(pkg, tmp_path):
    """the step in front of the on-disk format when a graph comes as pairs (OGB edge_index, a text file): edge list ->
    adjacency (both directions, duplicates collapsed, self-pairs dropped) -> prepare_dataset (padding, self-loops) -> the
    reference's files -> read back (test/data/prep.py:128-146 does this through DGL)."""
    ds = pkg.datasets
    src = np.array([0, 1, 1, 2, 4, 4, 0, 3], dtype=np.int64)
    dst = np.array([1, 0, 2, 2, 0, 0, 4, 5], dtype=np.int64)               # (1,0) repeats (0,1); (2,2) is a self-pair; (4,0) twice
    A = ds.adjacency_from_edges(src, dst, n=6)
    want = np.zeros((6, 6), np.float32)
    for u, v in [(0, 1), (1, 2), (0, 4), (3, 5)]:
        want[u, v] = want[v, u] = 1.0
    np.testing.assert_array_equal(A.toarray(), want)
    D = ds.adjacency_from_edges(src, dst, n=6, symmetric=False).toarray()
    assert D[4, 0] == 1.0 and D[0, 4] == 1.0 and D[3, 5] == 1.0 and D[5, 3] == 0.0 and D[2, 2] == 0.0
    with pytest.raises(ValueError):
        ds.adjacency_from_edges([0, 7], [1, 2], n=6)
    # text and .npy forms of the same list
    txt = tmp_path / "edges.txt"
    txt.write_text("# u v\n" + "\n".join(f"{u} {v}" for u, v in zip(src, dst)) + "\n")
    np.save(tmp_path / "edges.npy", np.stack([src, dst]))                   # [2, E], the OGB layout
    for path in (str(txt), str(tmp_path / "edges.npy")):
        s2, d2 = ds.read_edge_list(path)
        np.testing.assert_array_equal(s2, src)
        np.testing.assert_array_equal(d2, dst)
    X = np.arange(18, dtype=np.float32).reshape(6, 3)
    Y = np.array([0, 1, 2, 0, 1, 2])
    out = ds.prepare_dataset(str(tmp_path / "g"), A, X, Y, P=8, seed=0)
    (ip, ix, dv, n, m), X2, Y2, _ = ds.read_dataset(out)
    assert (n, m) == (8, 8) and X2.shape == (8, 8) and len(ix) == 8 + 8      # padded to a multiple of P; 8 edges + 8 self-loops
    dense = np.zeros((8, 8), np.float32)
    for r in range(8):
        dense[r, ix[ip[r]:ip[r + 1]]] = dv[ip[r]:ip[r + 1]]
    full = np.zeros((8, 8), np.float32)
    full[:6, :6] = want
    np.testing.assert_array_equal(dense, full + np.eye(8, dtype=np.float32))
    np.testing.assert_array_equal(X2[:6, :3], X)
    np.testing.assert_array_equal(Y2[:6, 0], Y)


def test_prep_command_line(pkg, tmp_path):
    """mg-gcn_amd/prep.py: edge list + .npy features / labels -> the files `mg_gcn train <dir>` reads (the role of the
    reference's test/data/prep.py, without DGL / OGB / network)."""
    import subprocess
    import sys
    rng = np.random.default_rng(0)
    n, E = 50, 300
    e = rng.integers(0, n, size=(E, 2))
    np.save(tmp_path / "e.npy", e)
    np.save(tmp_path / "x.npy", rng.standard_normal((n, 5)).astype(np.float32))
    np.save(tmp_path / "y.npy", rng.integers(0, 4, size=n))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "mg-gcn_amd", "prep.py"), "--edges", str(tmp_path / "e.npy"),
                        "--features", str(tmp_path / "x.npy"), "--labels", str(tmp_path / "y.npy"), "--out", str(tmp_path / "data" / "g"),
                        "-P", "4", "--seed", "3"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = r.stdout.splitlines()[0]
    assert out == str(tmp_path / "data" / "permuted" / "g") and "halo rows per exchange" in r.stdout
    (ip, ix, dv, nn, m), X, Y, S = pkg.datasets.read_dataset(out)
    assert nn == m == 52 and X.shape == (52, 8) and Y.shape == (52, 1)           # padded to multiples of P = 4
    A = pkg.datasets.adjacency_from_edges(e[:, 0], e[:, 1], n=n)
    assert len(ix) == A.nnz + 52                                                # + one self-loop per (padded) vertex
    rows = np.repeat(np.arange(52), np.diff(ip.astype(np.int64)))
    assert set(zip(rows.tolist(), ix.tolist())) == set(zip(ix.tolist(), rows.tolist()))   # pattern stays symmetric


def test_cmake_front_builds_the_cli(tmp_path):
    """mg-gcn_amd/CMakeLists.txt: the reference's target names (`mg_gcn_lib`, `mg_gcn`, option LOG; src/CMakeLists.txt:10-37) over
    the prebuilt C-ABI libraries -- configures, builds and the binary answers -h without a GPU (with -DLOG=ON: the scope timer of
    src/matrix.hpp:160-187 compiles in)."""
    import shutil
    if shutil.which("cmake") is None:
        pytest.skip("no cmake")
    b = tmp_path / "build"
    r = subprocess.run(["cmake", "-S", os.path.join(ROOT, "mg-gcn_amd"), "-B", str(b), "-DLOG=ON"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    r = subprocess.run(["cmake", "--build", str(b), "-j", "4"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    r = subprocess.run([str(b / "mg_gcn"), "-h"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "Usage" in r.stdout
