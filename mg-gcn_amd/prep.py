#!/usr/bin/env python3
"""Command-line front of datasets.prepare_dataset: what test/data/prep.py does for the reference (there through DGL / OGB and
the network), for graphs that are already on disk as an edge list -- the step in front of `mg_gcn train <dir> ...`.

    python mg-gcn_amd/prep.py --edges edges.txt|edge_index.npy --features X.npy --labels y.npy [--sets s.npy]
                              --out test/data/mygraph [-P 8] [--seed 1] [--partition blocks|perm.txt] [--directed]

Writes graph.bin / features.bin / labels.bin / sets.bin in the reference's formats (PIGO-CSR-v2 + dense u32-headed files,
prep.py:46-99): vertex count and feature width padded to multiples of P, a self-loop on every vertex, and -- seed != 0 -- one
random symmetric permutation, under <out>'s `permuted/` sibling like prep.py:80-94; `--partition` writes under `partitioned/`
instead (contiguous blocks p[i] = i n / P are then the parts: the hook prep.py:232-272 sketches with PaToH).  Prints the
directory and the halo volume of the written partition (rows each GPU would pull per exchange: prep.py:237-244's figure)."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--edges", required=True, help='text file of "u v" lines, or .npy of shape [E, 2] / [2, E]')
    ap.add_argument("--features", required=True, help=".npy float array [n, F]")
    ap.add_argument("--labels", required=True, help=".npy integer array [n]")
    ap.add_argument("--sets", default=None, help=".npy integer array [n]: 0 train, 1 validation, 2 test")
    ap.add_argument("--out", required=True)
    ap.add_argument("-P", type=int, default=8, help="GPUs the files must be divisible for (padding granularity)")
    ap.add_argument("--seed", type=int, default=1, help="random symmetric permutation (0: none)")
    ap.add_argument("--partition", default=None, help='"blocks" (datasets.partition_blocks) or a permutation file')
    ap.add_argument("--directed", action="store_true", help="keep the edge directions (default: both directions, like prep.py:137)")
    a = ap.parse_args(argv)
    ds = ge.load_package().datasets
    src, dst = ds.read_edge_list(a.edges)
    X = np.load(a.features, allow_pickle=False)
    y = np.load(a.labels, allow_pickle=False).reshape(-1)
    sets = np.load(a.sets, allow_pickle=False).reshape(-1) if a.sets else None
    if X.shape[0] != y.shape[0]:
        sys.exit(f"features hold {X.shape[0]} vertices, labels {y.shape[0]}")
    A = ds.adjacency_from_edges(src, dst, n=X.shape[0], symmetric=not a.directed)
    kw = {}
    if a.partition == "blocks":
        kw["partitioner"] = "blocks"
    elif a.partition:
        kw["permutation"] = a.partition
    out = ds.prepare_dataset(a.out, A, X, y, sets, P=a.P, seed=a.seed, **kw)
    ip, ix, _, n, _ = ds.read_csr(os.path.join(out, "graph.bin"))
    vol = ds.comm_volume_matrix(ip, ix, a.P)
    print(out)
    halo = int(vol.sum() - np.trace(vol))                              # off-diagonal: rows pulled from OTHER GPUs
    print(f"n = {n}, nnz = {len(ix)}, halo rows per exchange: {halo} of {(a.P - 1) * n} (all-gather)")
    return 0


if __name__ == "__main__":
    sys.exit(main())
