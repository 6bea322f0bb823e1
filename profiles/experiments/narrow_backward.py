"""d = 41 SpMM with the BACKWARD matrix of the Reddit-shaped graph (A: power-law row degrees,
uniform columns -- the forward matrix A^T has even rows and power-law column popularity):
forced lanes-per-entry x panel size, against what the plan picks on its own."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as g
pkg = g.load_package()
(ip, ix, dv), _, _ = pkg.datasets.synth_reddit_like(1.0, seed=1)
n = ip.shape[0] - 1
A = pkg.csr_matrix(ip, ix, dv, n); A.normalize(True)
mats = {"A (backward)": A, "A^T (forward)": A.transpose()}
ctx = pkg.context(0)
d = int(os.environ.get("EXP_D", "41"))
B = pkg.dn_matrix.from_numpy(np.random.default_rng(0).standard_normal((n, d), dtype=np.float32))
C = pkg.dn_matrix(n, d)
for name, M in mats.items():
    for lpe, panel in [(0, 4096), (0, 6144), (0, 8192), (0, 16384), (12, 8192), (12, 16384), (12, 32768), (16, 8192), (16, 16384), (16, 32768)]:
        if lpe: os.environ["MGGCN_SPMM_NARROW_LPE"] = str(lpe)
        else: os.environ.pop("MGGCN_SPMM_NARROW_LPE", None)
        os.environ["MGGCN_SPMM_PANEL_ROWS_NARROW"] = str(panel)
        buf = pkg.get_matmul_buffer(ctx, M, B, C)
        for _ in range(2): pkg.matmul(ctx, M, B, C, buf, 1.0, 0.0)
        ctx.sync(); ctx.record("a", 0)
        for _ in range(5): pkg.matmul(ctx, M, B, C, buf, 1.0, 0.0)
        ctx.record("b", 0); ctx.sync(); ctx.register_timer("t", "a", "b")
        print(f"{name} d={d} lpe={'auto' if not lpe else lpe} panel={panel}: {ctx.measure('t')/5:.3f} ms  plan {buf.nbytes()/1e9:.2f} GB "
              f"tasks {buf.num_sweep_tasks()} split rows {buf.num_split_rows()}", flush=True)
        del buf
