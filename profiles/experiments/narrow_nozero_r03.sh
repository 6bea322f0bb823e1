cd $GRAFT_REPO_ROOT
timeout -k 10 500 python3 -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "sweep or random or spmm" 2>&1 | tail -n 3
for SYM in 0 1; do
echo "== symmetric=$SYM"
SPMM_AB_SYMMETRIC=$SYM timeout -k 10 200 python3 profiles/experiments/spmm_ab.py "fast+nozero" 2>&1 | grep "nozero" | cut -c1-200
done
