cd $GRAFT_REPO_ROOT
for K in 0 64 128 256 4096; do
echo "== MGGCN_SPMM_HOT_EXPERIMENT=$K"
MGGCN_SPMM_HOT_EXPERIMENT=$K SPMM_AB_SYMMETRIC=1 python3 profiles/experiments/spmm_ab.py "hot=$K" 2>&1 | grep -E "hot experiment|hot=" | sort | uniq | head -6
done
