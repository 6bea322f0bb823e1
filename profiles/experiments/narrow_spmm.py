"""d = 41 SpMM on the Reddit shape: one-column-per-lane sweep vs the float4 quad form
(MGGCN_SPMM_SWEEP_QUAD=0/1) for a few panel sizes.  Usage: python profiles/experiments/narrow_spmm.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as g
pkg = g.load_package()
import torch

(ip, ix, dv), _, _ = pkg.datasets.synth_reddit_like(1.0, seed=1)
n = ip.shape[0] - 1
A = pkg.csr_matrix(ip, ix, dv, n); A.normalize(True); A = A.transpose()
ctx = pkg.context(0)
for d in (41, 44, 64, 16):
    B = pkg.dn_matrix.from_numpy(np.random.default_rng(0).standard_normal((n, d), dtype=np.float32))
    C = pkg.dn_matrix(n, d)
    ref = None
    for quad, panel, bpc in [(0, 0, 4), (1, 8192, 4), (1, 16384, 4), (1, 32768, 4)]:
        os.environ["MGGCN_SPMM_SWEEP_QUAD"] = str(quad)
        if panel: os.environ["MGGCN_SPMM_PANEL_ROWS_NARROW"] = str(panel)
        os.environ["MGGCN_SPMM_SWEEP_BLOCKS_PER_CU"] = str(bpc)
        t0 = time.time()
        buf = pkg.get_matmul_buffer(ctx, A, B, C) if quad else pkg.ops.spmm_buffer(
            ctx.lib, ctx.lib.mggcn_spmm_plan_create(A.n(), A.m(), A.indptr.ctypes.data, A.indices.ctypes.data, A.data.ctypes.data, 128))
        tb = time.time() - t0
        for _ in range(3): pkg.matmul(ctx, A, B, C, buf, 1.0, 0.0)
        ctx.sync()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s = torch.cuda.ExternalStream(ctx.stream(0))
        e0.record(s)
        for _ in range(10): pkg.matmul(ctx, A, B, C, buf, 1.0, 0.0)
        e1.record(s); ctx.sync()
        out = C.numpy()
        if ref is None: ref = out
        err = float(np.abs(out - ref).max() / np.abs(ref).max())
        print(f"d={d} quad={quad} panel={panel} blocks/CU={bpc}: {e0.elapsed_time(e1)/10:.3f} ms  plan {tb:.1f}s {buf.nbytes()/1e9:.2f} GB tasks {buf.num_sweep_tasks()} maxrel {err:.1e}", flush=True)
        del buf
