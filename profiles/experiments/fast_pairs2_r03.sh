cd $GRAFT_REPO_ROOT
timeout -k 10 300 python3 -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "sweep" 2>&1 | tail -n 2
for V in 0 1 0 1; do
echo "== MGGCN_SPMM_DEBUG_ONE_WAIT=$V (rotation by down-counter in both; symmetric stand-in)"
SPMM_AB_SYMMETRIC=1 MGGCN_SPMM_DEBUG_ONE_WAIT=$V timeout -k 10 200 python3 profiles/experiments/spmm_ab.py "one_wait=$V" 2>&1 | grep "one_wait=" | cut -c1-110
done
