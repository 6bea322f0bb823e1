cd $GRAFT_REPO_ROOT
timeout -k 10 300 python3 profiles/experiments/spmm_fuzz_r03.py 600 5 2>&1 | tail -n 12 | cut -c1-600
for L in 16 12; do
for V in 0 1; do
echo "== symmetric=1 MGGCN_SPMM_NARROW_LPE=$L MGGCN_SPMM_SPLIT_GROUPS=$V"
SPMM_AB_SYMMETRIC=1 MGGCN_SPMM_NARROW_LPE=$L MGGCN_SPMM_SPLIT_GROUPS=$V MGGCN_SPMM_PLAN_LOG=1 timeout -k 10 200 python3 profiles/experiments/spmm_ab.py "lpe=$L split_groups=$V" 2>&1 | grep -v amdgpu.ids | cut -c1-400 | grep "split_groups=\|form=sweep-narrow" | sed 's/.*| tasks/| tasks/'
done
done
