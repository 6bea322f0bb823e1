# VERDICT r03 item 8 (d = 41): five waves per SIMD at a smaller accumulator set -- eight rows per task, accumulator planes in
# v[64:95], 96 VGPRs (spmm_sweep_quad_lds_kernel<LPE, true>, MGGCN_SPMM_NARROW8=1), five workgroups per CU.
cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" python3 profiles/experiments/spmm_ab.py "$*" 2>/dev/null | tail -1; }
run A=default
run MGGCN_SPMM_SWEEP_ROWS_PER_TASK=8
run MGGCN_SPMM_SWEEP_ROWS_PER_TASK=8 MGGCN_SPMM_NARROW8=1
run MGGCN_SPMM_SWEEP_ROWS_PER_TASK=8 MGGCN_SPMM_NARROW8=1 MGGCN_SPMM_SWEEP_BLOCKS_PER_CU=5
run MGGCN_SPMM_SWEEP_ROWS_PER_TASK=8 MGGCN_SPMM_NARROW8=1 MGGCN_SPMM_SWEEP_BLOCKS_PER_CU=5 MGGCN_SPMM_PANEL_ROWS_NARROW=6144
run MGGCN_SPMM_SWEEP_ROWS_PER_TASK=8 MGGCN_SPMM_NARROW8=1 MGGCN_SPMM_SWEEP_BLOCKS_PER_CU=5 MGGCN_SPMM_PANEL_ROWS_NARROW=12288
echo "== parity of the variant"
MGGCN_SPMM_SWEEP_ROWS_PER_TASK=8 MGGCN_SPMM_NARROW8=1 MGGCN_SPMM_SWEEP_BLOCKS_PER_CU=5 timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -q -x -k "narrow or sweep" 2>&1 | tail -3
