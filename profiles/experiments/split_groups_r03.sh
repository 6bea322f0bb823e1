# split form of the NARROW entry stream (one part per lane group, no run padding) against the padded-run form
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python3 -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "sweep or random or spmm" 2>&1 | tail -n 3
for SYM in 0 1; do
for V in 0 1; do
echo "== symmetric=$SYM MGGCN_SPMM_SPLIT_GROUPS=$V"
SPMM_AB_SYMMETRIC=$SYM MGGCN_SPMM_SPLIT_GROUPS=$V MGGCN_SPMM_PLAN_LOG=1 timeout -k 10 200 python3 profiles/experiments/spmm_ab.py "split_groups=$V" 2>&1 | grep -v amdgpu.ids | cut -c1-400 | grep "split_groups=\|form=sweep-narrow"
done
done
