# split form of the wide entry stream (no run padding) against the padded-run form: parity first, then time
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python3 -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "sweep or random" 2>&1 | tail -n 4
for SYM in 0 1; do
for V in 0 1; do
echo "== symmetric=$SYM MGGCN_SPMM_SPLIT_PAIRS=$V"
SPMM_AB_SYMMETRIC=$SYM MGGCN_SPMM_SPLIT_PAIRS=$V MGGCN_SPMM_PLAN_LOG=1 timeout -k 10 200 python3 profiles/experiments/spmm_ab.py "split=$V" 2>&1 | grep -v amdgpu.ids | cut -c1-420 | grep "split=\|form=sweep " 
done
done
