# round 3: the hot-row form of the d = 128 kernel (MGGCN_SPMM_HOT_ROWS = K rows of every panel in LDS), symmetric stand-in
cd $GRAFT_REPO_ROOT
for K in 0 32 64 128; do
echo "== MGGCN_SPMM_HOT_ROWS=$K"
MGGCN_SPMM_HOT_ROWS=$K SPMM_AB_SYMMETRIC=1 timeout -k 10 200 python3 profiles/experiments/spmm_ab.py "hot_rows=$K" 2>&1 | tail -1
done
