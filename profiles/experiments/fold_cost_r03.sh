# what the run folds cost today: the same kernels with every fold skipped (results wrong, time only)
cd $GRAFT_REPO_ROOT
for V in 0 1; do
echo "== MGGCN_SPMM_DEBUG_SKIP_FOLD=$V (symmetric stand-in)"
SPMM_AB_SYMMETRIC=1 MGGCN_SPMM_DEBUG_SKIP_FOLD=$V timeout -k 10 200 python3 profiles/experiments/spmm_ab.py "skip_fold=$V" 2>&1 | grep "skip_fold=" | cut -c1-200
done
