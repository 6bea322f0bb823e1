"""On-disk dataset format and synthetic graph generators (host side, numpy).

The format is the one the reference's data-prep script writes and its binary
reads (reference: test/data/prep.py:46-76 writer, :198-209 reader;
src/matrix.hpp:224-234 and :486-492 readers; src/main.cpp:82-85 file names):

``graph.bin``     ASCII ``PIGO-CSR-v2`` | u8 4 | u8 4 | u32 n | u32 nnz | u32 nrows |
                  u32 ncols | u32 indptr[nrows+1] | u32 indices[nnz] | f32 data[nnz]
``features.bin``  u32 N | u32 M | f32 payload, row-major
``labels.bin``    u32 N | u32 1 | u32 payload (read back as int32)
``sets.bin``      u32 N | u32 1 | u32 payload (0 train / 1 val / 2 test)

Only 4-byte index / offset widths are supported, as in the reference
(``x_t = v_t = unsigned``, src/main.cpp:43-45).
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import numpy as np

MAGIC = b"PIGO-CSR-v2"


class format_error(RuntimeError):
    """Mirrors matrix_error (reference src/matrix.hpp:32-37): unsupported / malformed file."""


def write_csr(path: str, indptr, indices, data, n_cols: Optional[int] = None) -> None:
    indptr = np.ascontiguousarray(indptr, dtype="<u4")
    indices = np.ascontiguousarray(indices, dtype="<u4")
    data = np.ascontiguousarray(data, dtype="<f4")
    n = indptr.shape[0] - 1
    nnz = int(indptr[-1])
    if n_cols is None:
        n_cols = n
    if indices.shape[0] != nnz or data.shape[0] != nnz:
        raise format_error("indptr[-1] does not match indices/data length")
    with open(path, "wb") as f:
        f.write(MAGIC)
        f.write(np.array([4, 4], dtype=np.uint8).tobytes())
        f.write(np.array([n, nnz, n, n_cols], dtype="<u4").tobytes())
        f.write(indptr.tobytes())
        f.write(indices.tobytes())
        f.write(data.tobytes())


def read_csr(path: str) -> Tuple[np.ndarray, np.ndarray, np.ndarray, int, int]:
    """Returns (indptr u32[n+1], indices u32[nnz], data f32[nnz], n_rows, n_cols)."""
    if not str(path).endswith(".bin"):
        raise format_error("File type is not supported.")      # reference src/matrix.hpp:282
    with open(path, "rb") as f:
        head = f.read(13)
        if len(head) != 13 or head[:11] != MAGIC:
            raise format_error(f"{path}: not a PIGO-CSR-v2 file")
        if head[11] != 4 or head[12] != 4:
            raise format_error(f"{path}: only 4-byte index/offset widths are supported")
        hdr = np.frombuffer(f.read(16), dtype="<u4")
        if hdr.shape[0] != 4:
            raise format_error(f"{path}: truncated header")
        _n, nnz, nrows, ncols = (int(x) for x in hdr)
        indptr = np.fromfile(f, dtype="<u4", count=nrows + 1)
        indices = np.fromfile(f, dtype="<u4", count=nnz)
        data = np.fromfile(f, dtype="<f4", count=nnz)
    if indptr.shape[0] != nrows + 1 or indices.shape[0] != nnz or data.shape[0] != nnz:
        raise format_error(f"{path}: truncated payload")
    if nnz and (int(indptr[-1]) - int(indptr[0]) != nnz):
        raise format_error(f"{path}: indptr does not cover nnz")
    return indptr, indices, data, nrows, ncols


def write_dense(path: str, A, dtype) -> None:
    A = np.ascontiguousarray(A, dtype=dtype)
    if A.ndim == 1:
        A = A.reshape(-1, 1)
    with open(path, "wb") as f:
        f.write(np.array(A.shape, dtype="<u4").tobytes())
        f.write(A.tobytes())


def read_dense(path: str, dtype) -> np.ndarray:
    if not str(path).endswith(".bin"):
        raise format_error("File type is not supported.")      # reference src/matrix.hpp:518
    with open(path, "rb") as f:
        shape = np.frombuffer(f.read(8), dtype="<u4")
        if shape.shape[0] != 2:
            raise format_error(f"{path}: truncated header")
        n, m = int(shape[0]), int(shape[1])
        payload = np.fromfile(f, dtype=dtype, count=n * m)
    if payload.shape[0] != n * m:
        raise format_error(f"{path}: truncated payload")
    return payload.reshape(n, m)


def write_dataset(dirname: str, indptr, indices, data, features, labels, sets=None) -> None:
    """Writes graph.bin / features.bin / labels.bin / sets.bin (reference prep.py:78-99)."""
    os.makedirs(dirname, exist_ok=True)
    write_csr(os.path.join(dirname, "graph.bin"), indptr, indices, data)
    write_dense(os.path.join(dirname, "features.bin"), features, "<f4")
    labels = np.asarray(labels).reshape(-1, 1)
    write_dense(os.path.join(dirname, "labels.bin"), labels, "<u4")
    if sets is None:
        sets = np.zeros_like(labels)
    write_dense(os.path.join(dirname, "sets.bin"), np.asarray(sets).reshape(-1, 1), "<u4")


def read_dataset(dirname: str):
    """Returns ((indptr, indices, data, n, m), X f32[n,F], Y int32[n,1], S int32[n,1])
    exactly as reference src/main.cpp:82-85 loads them."""
    g = read_csr(os.path.join(dirname, "graph.bin"))
    X = read_dense(os.path.join(dirname, "features.bin"), "<f4")
    Y = read_dense(os.path.join(dirname, "labels.bin"), "<i4")
    S = read_dense(os.path.join(dirname, "sets.bin"), "<i4")
    return g, X, Y, S


# ----------------------------------------------------------------------------
# Synthetic inputs.  Deterministic, numpy only.
# ----------------------------------------------------------------------------
def synth_uniform_csr(n: int, deg: int, seed: int = 0):
    """C1: n rows, exactly ``deg`` distinct random columns per row (nnz = n*deg),
    values uniform(0,1).  n=10_000, deg=10 is BASELINE.json configs[0]."""
    rng = np.random.default_rng(seed)
    cols = np.empty((n, deg), dtype=np.uint32)
    # distinct columns per row: draw with a stride trick, then fix the rare collisions
    cols[:] = rng.integers(0, n, size=(n, deg), dtype=np.uint32)
    cols.sort(axis=1)
    dup = (np.diff(cols, axis=1) == 0)
    while dup.any():
        r, c = np.nonzero(dup)
        cols[r, c + 1] = rng.integers(0, n, size=r.shape[0], dtype=np.uint32)
        cols.sort(axis=1)
        dup = (np.diff(cols, axis=1) == 0)
    indptr = (np.arange(n + 1, dtype=np.uint64) * deg).astype(np.uint32)
    data = rng.random(n * deg, dtype=np.float32)
    return indptr, cols.reshape(-1), data


def synth_powerlaw_csr(n: int, nnz_target: int, max_deg: int, seed: int = 1, alpha: float = 1.3,
                       self_loops: bool = True):
    """Reddit-shaped stand-in: heavy-tailed out-degrees with the requested
    total, uniformly random columns (the reference trains on a randomly permuted graph,
    prep.py:87-94, so columns carry no locality), a self-loop on every row (prep.py:113),
    unit values (the trainer normalises them).  Duplicate columns inside a row are allowed
    (they are legal CSR and every consumer handles them); indices are NOT sorted."""
    if not (n <= nnz_target <= n * max_deg):
        raise ValueError(f"cannot place {nnz_target} non-zeros in {n} rows of 1..{max_deg} entries")
    rng = np.random.default_rng(seed)
    # Pareto-like degrees clipped to [1, max_deg], rescaled to hit the nnz target exactly
    raw = (rng.pareto(alpha, size=n) + 1.0)
    w = raw
    for _ in range(64):          # water-filling: rescale, clip, repeat until the total fits
        w = np.clip(w * (nnz_target / w.sum()), 1.0, float(max_deg))
        if abs(w.sum() - nnz_target) < 0.5 * n:
            break
    deg = np.clip(np.floor(w), 1, max_deg).astype(np.int64)
    diff = int(nnz_target - deg.sum())
    # distribute the remainder (or remove the surplus) over random rows, one entry each
    while diff != 0:
        k = min(abs(diff), n)
        rows = rng.choice(n, size=k, replace=False)
        if diff > 0:
            ok = rows[deg[rows] < max_deg]
            deg[ok] += 1
            diff -= ok.shape[0]
        else:
            ok = rows[deg[rows] > 1]
            deg[ok] -= 1
            diff += ok.shape[0]
    indptr = np.zeros(n + 1, dtype=np.uint64)
    np.cumsum(deg, out=indptr[1:])
    assert indptr[-1] == nnz_target and indptr[-1] < 2 ** 32
    indices = rng.integers(0, n, size=int(nnz_target), dtype=np.uint32)
    if self_loops:
        indices[indptr[:-1].astype(np.int64)] = np.arange(n, dtype=np.uint32)
    data = np.ones(int(nnz_target), dtype=np.float32)
    return indptr.astype(np.uint32), indices, data


def synth_symmetric_powerlaw_csr(n: int, nnz_target: int, max_deg: int, seed: int = 1, alpha: float = 1.3):
    """Reddit's STRUCTURE, not only its shape: an undirected simple graph stored in both directions plus a
    self-loop on every vertex -- the reference's Reddit is exactly that: 114 615 892 directed entries (two per
    undirected edge) + 232 968 self-loops = the 114 848 860 of test/test_matrix.cpp:48-58 (prep.py:113 adds the
    loops).  Pattern A = A^T, so the forward matrix (A D^-1)^T and the backward matrix A D^-1 BOTH have
    power-law rows AND popular columns (synth_powerlaw_csr gives each matrix only one of the two).
      * expected degrees: the same clipped-Pareto sequence as synth_powerlaw_csr;
      * edges: endpoints drawn in proportion to the expected degree (Chung-Lu), self-pairs and repeated pairs
        dropped, redrawn until EXACTLY (nnz_target - n) / 2 distinct pairs exist -- a simple graph;
      * rows hold their columns in ascending order (lower neighbours, self-loop, upper neighbours), as a
        scipy-written graph.bin does (prep.py:46-76 writes `g.adj(scipy_fmt='csr')`); unit values.
    Deterministic for a seed.  nnz_target - n must be even."""
    m2 = nnz_target - n
    if m2 < 0 or m2 % 2 or m2 // 2 > n * (n - 1) // 2:
        raise ValueError(f"{nnz_target} non-zeros cannot be n = {n} self-loops plus both directions of a simple graph")
    m = m2 // 2
    rng = np.random.default_rng(seed)
    raw = rng.pareto(alpha, size=n) + 1.0
    w = raw
    for _ in range(64):
        w = np.clip(w * (m2 / max(w.sum(), 1.0)), 1.0, float(max(1, max_deg - 1)))
        if abs(w.sum() - m2) < 0.5 * n:
            break
    # stub table: vertex i appears round(w_i) times; a uniform draw from it is a draw ~ expected degree
    stubs = np.repeat(np.arange(n, dtype=np.uint32), np.maximum(1, np.rint(w)).astype(np.int64))
    T = stubs.shape[0]

    def draw(k):
        """k vertex pairs (u < v) as keys u*n + v, endpoints ~ expected degree, self-pairs dropped"""
        a = stubs[rng.integers(0, T, size=k)].astype(np.int64)
        b = stubs[rng.integers(0, T, size=k)].astype(np.int64)
        lo, hi = np.minimum(a, b), np.maximum(a, b)
        keep = lo != hi
        return lo[keep] * n + hi[keep]

    keys = np.empty(0, dtype=np.int64)
    while keys.shape[0] < m:
        need = m - keys.shape[0]
        keys = np.concatenate([keys, draw(int(need * 1.08) + 16)])
        keys.sort()
        first = np.ones(keys.shape[0], dtype=bool)
        np.not_equal(keys[1:], keys[:-1], out=first[1:])
        keys = keys[first]
        if keys.shape[0] > m:                                   # too many: drop a random subset, keep the order
            first = np.ones(keys.shape[0], dtype=bool)
            first[rng.choice(keys.shape[0], size=keys.shape[0] - m, replace=False)] = False
            keys = keys[first]
    del stubs, first
    u = keys // n                                               # sorted by (u, v): the upper triangle in CSR order
    v = (keys - u * n).astype(np.int32)
    del keys
    up_cnt = np.bincount(u, minlength=n).astype(np.int64)       # neighbours above the diagonal, per row
    del u
    up_ptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(up_cnt, out=up_ptr[1:])
    # lower triangle = the transpose of the upper one: scipy's CSR -> CSC is a stable counting sort (rows ascending
    # inside every column), O(m)
    import scipy.sparse as sp
    low = sp.csr_matrix((np.ones(m, dtype=np.int8), v, up_ptr), shape=(n, n)).tocsc()
    lo_cnt = np.diff(low.indptr).astype(np.int64)
    indptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(lo_cnt + 1 + up_cnt, out=indptr[1:])
    assert indptr[-1] == nnz_target and indptr[-1] < 2 ** 32
    indices = np.empty(nnz_target, dtype=np.uint32)
    pos = np.arange(m, dtype=np.int64)
    # row r = [lower neighbours | r | upper neighbours]
    indices[pos + np.repeat(indptr[:-1] + lo_cnt + 1 - up_ptr[:-1], up_cnt)] = v
    indices[pos + np.repeat(indptr[:-1] - low.indptr[:-1].astype(np.int64), lo_cnt)] = low.indices
    del pos, v, low
    indices[indptr[:-1] + lo_cnt] = np.arange(n, dtype=np.uint32)
    return indptr.astype(np.uint32), indices, np.ones(nnz_target, dtype=np.float32)


REDDIT_SHAPE = dict(n=232_968, nnz=114_848_860, features=608, classes=41, max_deg=21_657)


def synth_reddit_like(scale: float = 1.0, seed: int = 1, symmetric: bool = False):
    """Graph + features + labels with Reddit's published shape (reference
    test/test_matrix.cpp:45-62: n=232 968, nnz=114 848 860, F=608; 41 classes), or a
    ``scale``d-down version with the same mean degree (n multiple of 8 as prep.py:101-103).
    ``symmetric``: pattern A = A^T like the real dataset (synth_symmetric_powerlaw_csr); the default is the
    SURVEY.md 8(d) stand-in the headline numbers are quoted on (random columns)."""
    n = int(REDDIT_SHAPE["n"] * scale) // 8 * 8
    nnz = int(REDDIT_SHAPE["nnz"] * scale)
    mean_deg = nnz / max(n, 1)
    # keep the published maximum at full scale; scaled-down graphs keep the skew ratio
    # (max/mean ~ 44) as far as duplicates-allowed rows make sense
    max_deg = int(max(4 * mean_deg + 8, REDDIT_SHAPE["max_deg"] * min(1.0, scale * 4)))
    if symmetric:
        nnz -= (nnz - n) % 2
        indptr, indices, data = synth_symmetric_powerlaw_csr(n, nnz, min(max_deg, n - 1), seed)
    else:
        indptr, indices, data = synth_powerlaw_csr(n, nnz, max_deg, seed)
    rng = np.random.default_rng(seed + 1)
    X = rng.standard_normal((n, REDDIT_SHAPE["features"]), dtype=np.float32)
    Y = rng.integers(0, REDDIT_SHAPE["classes"], size=(n, 1)).astype(np.int32)
    Y[0, 0] = REDDIT_SHAPE["classes"] - 1          # num_labels = 1 + max(Y) (main.cpp:89)
    return (indptr, indices, data), X, Y


PRODUCTS_SHAPE = dict(n=2_449_032, nnz=126_200_000, features=128, classes=47, max_deg=17_500)


def synth_products_like(scale: float = 1.0, seed: int = 5, symmetric: bool = True):
    """ogbn-products' public shape (BASELINE.json configs[3]; SURVEY.md 8: n = 2 449 032 after padding to a multiple
    of 8, ~126.2 M non-zeros with both directions + self-loops, mean degree 51; features 100 padded to 128 as
    prep.py:122-124 does, 47 classes -> 48 at P = 8): B = n x 128 floats = 1.25 GB >> Infinity Cache, so the SpMM is
    HBM-bound here.  The OGB graph is undirected, hence symmetric by default."""
    n = int(PRODUCTS_SHAPE["n"] * scale) // 8 * 8
    nnz = int(PRODUCTS_SHAPE["nnz"] * scale)
    max_deg = int(max(4 * nnz / max(n, 1) + 8, PRODUCTS_SHAPE["max_deg"] * min(1.0, scale * 4)))
    if symmetric:
        nnz -= (nnz - n) % 2
        indptr, indices, data = synth_symmetric_powerlaw_csr(n, nnz, min(max_deg, n - 1), seed)
    else:
        indptr, indices, data = synth_powerlaw_csr(n, nnz, max_deg, seed)
    rng = np.random.default_rng(seed + 1)
    X = rng.standard_normal((n, PRODUCTS_SHAPE["features"]), dtype=np.float32)
    Y = rng.integers(0, PRODUCTS_SHAPE["classes"], size=(n, 1)).astype(np.int32)
    Y[0, 0] = PRODUCTS_SHAPE["classes"] - 1
    return (indptr, indices, data), X, Y


# ----------------------------------------------------------------------------
# Data preparation without DGL: what the reference's
# test/data/prep.py does to a graph before writing it (serialize_dgl_graph :100-126 and
# serialize_dataset :78-99), on plain numpy / scipy inputs.
# ----------------------------------------------------------------------------
def read_csr_rows(path: str, row_begin: int, row_end: int):
    """Rows [row_begin, row_end) of a graph.bin WITHOUT reading the rest: header, the indptr slice, then one
    seek per array (the reference loads the whole file once per process, src/main.cpp:82; with one process per
    GPU that would be P whole copies on one host).  Returns (indptr u32[rows+1] re-based to 0, indices, data,
    n_rows_total, n_cols)."""
    if not str(path).endswith(".bin"):
        raise format_error("File type is not supported.")
    with open(path, "rb") as f:
        head = f.read(13)
        if len(head) != 13 or head[:11] != MAGIC or head[11] != 4 or head[12] != 4:
            raise format_error(f"{path}: not a 4-byte PIGO-CSR-v2 file")
        _n, nnz, nrows, ncols = (int(x) for x in np.frombuffer(f.read(16), dtype="<u4"))
        if not (0 <= row_begin <= row_end <= nrows):
            raise format_error(f"{path}: rows [{row_begin}, {row_end}) outside 0..{nrows}")
        base = 13 + 16
        f.seek(base + 4 * row_begin)
        ip = np.fromfile(f, dtype="<u4", count=row_end - row_begin + 1)
        lo, hi = int(ip[0]), int(ip[-1])
        f.seek(base + 4 * (nrows + 1) + 4 * lo)
        indices = np.fromfile(f, dtype="<u4", count=hi - lo)
        f.seek(base + 4 * (nrows + 1) + 4 * nnz + 4 * lo)
        data = np.fromfile(f, dtype="<f4", count=hi - lo)
    if ip.shape[0] != row_end - row_begin + 1 or indices.shape[0] != hi - lo or data.shape[0] != hi - lo:
        raise format_error(f"{path}: truncated payload")
    return (ip - np.uint32(lo)).astype(np.uint32), indices, data, nrows, ncols


def read_dense_rows(path: str, dtype, row_begin: int, row_end: int) -> np.ndarray:
    """rows [row_begin, row_end) of a dense .bin file (features / labels / sets), one seek"""
    with open(path, "rb") as f:
        shape = np.frombuffer(f.read(8), dtype="<u4")
        n, m = int(shape[0]), int(shape[1])
        if not (0 <= row_begin <= row_end <= n):
            raise format_error(f"{path}: rows [{row_begin}, {row_end}) outside 0..{n}")
        f.seek(8 + np.dtype(dtype).itemsize * m * row_begin)
        out = np.fromfile(f, dtype=dtype, count=(row_end - row_begin) * m)
    if out.shape[0] != (row_end - row_begin) * m:
        raise format_error(f"{path}: truncated payload")
    return out.reshape(row_end - row_begin, m)


def dense_shape(path: str):
    with open(path, "rb") as f:
        shape = np.frombuffer(f.read(8), dtype="<u4")
    return int(shape[0]), int(shape[1])


# ----------------------------------------------------------------------------
# Partitioner hook in front of the 1D row partition.  The reference's data-prep script
# takes a vertex permutation from a file or from PaToH (test/data/prep.py:20, :232-247: `-p <permutation-file>`,
# `patoh(g, parts, 'RWS')` commented out) and then cuts the permuted graph into contiguous equal blocks
# p[i] = i*n/P (:253-256).  Same contract here: a partitioner is anything that returns a permutation; the blocks
# stay contiguous and equal, so the trainer needs no change.
# ----------------------------------------------------------------------------
def partition_blocks(adj, P: int, sweeps: int = 10, seed: int = 0) -> np.ndarray:
    """In-repo partitioner: P equal blocks by breadth-first growing + balanced label propagation.
    Returns ``perm`` with new vertex i = old vertex perm[i] (block k = perm[k*n/P:(k+1)*n/P]).
      1. seeds: breadth-first order from a pseudo-peripheral vertex, cut into P equal chunks;
      2. refinement sweeps (Kernighan-Lin style): every vertex counts its neighbours per block (one sparse
         product); for every pair of blocks (a, b) the vertices of a and of b are ranked by the gain of crossing
         and the best-ranked PAIRS with a positive total gain are SWAPPED -- block sizes never change, so
         n % P == 0 is preserved exactly (src/dist_matrix.hpp:428).
    Deterministic for a given seed.  PaToH-quality cuts are not the point: the hook is."""
    import scipy.sparse as sp
    from scipy.sparse.csgraph import breadth_first_order
    if isinstance(adj, tuple):
        ip, ix, dv = adj
        n0 = len(ip) - 1
        adj = sp.csr_matrix((np.ones(len(ix), dtype=np.float32), np.asarray(ix), np.asarray(ip)), shape=(n0, n0))
    A = sp.csr_matrix(adj)
    n = A.shape[0]
    if n % P != 0:
        raise ValueError(f"n = {n} is not a multiple of P = {P} (pad first: prepare_dataset does)")
    S = sp.csr_matrix((np.ones(A.nnz, dtype=np.float32), A.indices, A.indptr), shape=A.shape)
    S = sp.csr_matrix(S + S.T)                          # communication is symmetric: a cut edge costs both sides
    S.setdiag(0); S.eliminate_zeros()
    rng = np.random.default_rng(seed)
    # breadth-first order over all components, from a far vertex of a first sweep
    seen = np.zeros(n, dtype=bool)
    order = []
    start = int(rng.integers(0, n))
    far = breadth_first_order(S, start, directed=False, return_predecessors=False)[-1]
    for s0 in [int(far)] + list(range(n)):
        if seen[s0]:
            continue
        comp = breadth_first_order(S, s0, directed=False, return_predecessors=False)
        comp = comp[~seen[comp]]
        seen[comp] = True
        order.append(comp)
        if seen.all():
            break
    order = np.concatenate(order)
    size = n // P
    label = np.empty(n, dtype=np.int64)
    label[order] = np.arange(n) // size
    for _ in range(max(0, sweeps)):
        onehot = sp.csr_matrix((np.ones(n, dtype=np.float32), (np.arange(n), label)), shape=(n, P))
        cnt = np.asarray((S @ onehot).todense())                       # neighbours per block
        free = np.ones(n, dtype=bool)                                  # one move per vertex per sweep (stale counts)
        moved = 0
        for a in range(P):
            for b in range(a + 1, P):
                va = np.nonzero((label == a) & free)[0]
                vb = np.nonzero((label == b) & free)[0]
                ga = cnt[va, b] - cnt[va, a]                           # gain of a -> b
                gb = cnt[vb, a] - cnt[vb, b]                           # gain of b -> a
                oa, ob = np.argsort(-ga, kind="stable"), np.argsort(-gb, kind="stable")
                m = min(oa.size, ob.size)
                pair = ga[oa[:m]] + gb[ob[:m]]                         # Kernighan-Lin pair gain (sorted: decreasing)
                k = int(np.searchsorted(-pair, 0.0, side="left"))      # pairs with a positive total
                k = min(k, max(1, m // 2))
                if k == 0 or pair[0] <= 0:
                    continue
                sa, sb = va[oa[:k]], vb[ob[:k]]
                label[sa], label[sb] = b, a
                free[sa] = free[sb] = False
                moved += 2 * k
        if moved == 0:
            break
    return np.argsort(label, kind="stable").astype(np.int64)


def read_permutation_file(path: str) -> np.ndarray:
    """whitespace-separated vertex ids, the format test/data/prep.py:241-243 reads"""
    with open(path) as f:
        return np.array([int(t) for t in f.read().split()], dtype=np.int64)


def comm_volume_matrix(indptr, indices, P: int) -> np.ndarray:
    """L[i, j] = distinct columns of block j referenced by the rows of block i -- exactly the matrix
    test/data/prep.py:258-266 prints (diagonal included, as there)."""
    indptr = np.asarray(indptr, dtype=np.int64)
    n = indptr.shape[0] - 1
    p = [i * n // P for i in range(P + 1)]
    L = np.zeros((P, P), dtype=np.int64)
    for i in range(P):
        mask = np.zeros(n, dtype=bool)
        mask[np.asarray(indices[indptr[p[i]]:indptr[p[i + 1]]], dtype=np.int64)] = True
        for j in range(P):
            L[i, j] = int(mask[p[j]:p[j + 1]].sum())
    return L


def adjacency_from_edges(src, dst, n: Optional[int] = None, weights=None, symmetric: bool = True):
    """Edge list -> scipy CSR adjacency (the step in front of prepare_dataset when the graph comes as pairs, e.g. an OGB
    `edge_index` or a text file of "u v" lines; the reference goes through DGL for this, test/data/prep.py:128-146).
    ``symmetric``: both directions of every edge (prep.py builds its graphs undirected: dgl.to_bidirected, :137);
    repeated pairs collapse to ONE entry of weight 1 (or of the summed ``weights``; with ``symmetric`` the larger of the two
    directions), self-pairs are dropped -- prepare_dataset adds the self-loops (prep.py:113)."""
    import scipy.sparse as sp
    src = np.asarray(src, dtype=np.int64).reshape(-1)
    dst = np.asarray(dst, dtype=np.int64).reshape(-1)
    if src.shape != dst.shape:
        raise ValueError("src and dst must have the same length")
    if src.size and (src.min() < 0 or dst.min() < 0):
        raise ValueError("negative vertex id")
    n = int(max(src.max(initial=-1), dst.max(initial=-1)) + 1) if n is None else int(n)
    if src.size and max(src.max(), dst.max()) >= n:
        raise ValueError("vertex id out of range")
    w = np.ones(src.shape[0], dtype=np.float64) if weights is None else np.asarray(weights, dtype=np.float64).reshape(-1)
    keep = src != dst
    src, dst, w = src[keep], dst[keep], w[keep]
    A = sp.coo_matrix((w, (src, dst)), shape=(n, n)).tocsr()          # duplicates are summed here
    if symmetric:
        A = A.maximum(A.T)                                             # weighted: the larger of the two directions
    if weights is None:
        A.data[:] = 1.0
    A = sp.csr_matrix(A, dtype=np.float32)
    A.sort_indices()
    return A


def read_edge_list(path: str):
    """(src, dst) from a text file of "u v" lines ('#' comments; extra columns ignored) or from an .npy file holding an
    array of shape [E, 2] or [2, E] (the OGB `edge_index` layout)."""
    if path.endswith(".npy"):
        e = np.load(path, allow_pickle=False)
        if e.ndim != 2 or 2 not in e.shape:
            raise ValueError("edge array must be [E, 2] or [2, E]")
        if e.shape[1] == 2:                                            # [E, 2] (a 2 x 2 array is read this way)
            e = e.T
        return np.asarray(e[0], dtype=np.int64), np.asarray(e[1], dtype=np.int64)
    e = np.loadtxt(path, dtype=np.int64, comments="#", usecols=(0, 1), ndmin=2)
    return e[:, 0], e[:, 1]


def prepare_dataset(dirname: str, adj, features, labels, sets=None, P: int = 8, seed: int = 0,
                    permutation=None, partitioner=None) -> str:
    """adj: scipy sparse (n x n) or (indptr, indices, data) CSR triple; features [n x F];
    labels [n]; sets [n] in {0 train, 1 val, 2 test}.
      * pads the vertex count and the feature width to multiples of P with zero vertices /
        zero columns (prep.py:101-103, :122-124; padding vertices get label 0, set 0),
      * adds a self-loop to every vertex, padding included (prep.py:113),
      * seed != 0: applies one random symmetric permutation to graph, features, labels, sets
        and writes under <dirname>/permuted/... like prep.py:80-94,
      * ``permutation`` (vector over the PADDED vertices, or the path of a permutation file, prep.py:241-243) or
        ``partitioner`` ("blocks" = partition_blocks, or a callable (adj, P) -> permutation): the partitioner
        hook -- the permuted graph is written under <dirname>/partitioned/...; its contiguous blocks
        p[i] = i*n/P are the parts,
      * writes graph.bin / features.bin / labels.bin / sets.bin.  Returns the directory."""
    import scipy.sparse as sp
    if isinstance(adj, tuple):
        ip, ix, dv = adj
        n0 = len(ip) - 1
        adj = sp.csr_matrix((np.asarray(dv, dtype=np.float32), np.asarray(ix), np.asarray(ip)), shape=(n0, n0))
    adj = sp.csr_matrix(adj, dtype=np.float32)
    n0 = adj.shape[0]
    features = np.asarray(features, dtype=np.float32).reshape(n0, -1)
    labels = np.asarray(labels).reshape(n0).astype(np.int64)
    sets = np.zeros(n0, dtype=np.int64) if sets is None else np.asarray(sets).reshape(n0).astype(np.int64)
    n = (n0 + P - 1) // P * P
    F0 = features.shape[1]
    F = (F0 + P - 1) // P * P
    adj = sp.csr_matrix((adj.data, adj.indices, np.concatenate([adj.indptr, np.full(n - n0, adj.indptr[-1])])), shape=(n, n))
    adj = sp.csr_matrix(adj + sp.eye(n, dtype=np.float32, format="csr") - sp.diags(adj.diagonal(), format="csr"))
    adj.data[:] = np.where(adj.data != 0, adj.data, 1.0)
    feats = np.zeros((n, F), dtype=np.float32)
    feats[:n0, :F0] = features
    labs = np.zeros(n, dtype=np.int64); labs[:n0] = labels
    st = np.zeros(n, dtype=np.int64); st[:n0] = sets
    out = dirname
    perm = None
    head, tail = os.path.split(os.path.normpath(dirname))
    if permutation is not None or partitioner is not None:
        if permutation is not None:
            perm = read_permutation_file(permutation) if isinstance(permutation, (str, os.PathLike)) else np.asarray(permutation, dtype=np.int64)
        elif callable(partitioner):
            perm = np.asarray(partitioner(adj, P), dtype=np.int64)
        elif partitioner == "blocks":
            perm = partition_blocks(adj, P, seed=seed)
        else:
            raise ValueError(f"unknown partitioner {partitioner!r}")
        if perm.shape[0] != n or not np.array_equal(np.sort(perm), np.arange(n)):
            raise ValueError("the permutation must cover the padded vertex set exactly once")
        out = os.path.join(head, "partitioned", tail)
    elif seed != 0:
        out = os.path.join(head, "permuted", tail)
        perm = np.random.default_rng(seed).permutation(n)
    if perm is not None:
        adj = adj[perm][:, perm]
        feats, labs, st = feats[perm], labs[perm], st[perm]
    adj = sp.csr_matrix(adj)
    adj.sort_indices()
    write_dataset(out, adj.indptr, adj.indices, adj.data, feats, labs, st)
    return out
