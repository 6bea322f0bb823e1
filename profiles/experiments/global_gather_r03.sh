# the pair kernel's gathers as global_load_dwordx4 (SGPR base + VGPR offset) instead of buffer_load_dwordx4 ... soffset
cd $GRAFT_REPO_ROOT
timeout -k 10 200 python3 -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "sweep_spmm_widths" 2>&1 | tail -n 1
for i in 1 2; do
SPMM_AB_SYMMETRIC=1 timeout -k 10 200 python3 profiles/experiments/spmm_ab.py "global_gather" 2>&1 | grep "global_gather" | cut -c1-110
done
