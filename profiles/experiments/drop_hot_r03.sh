# premise test for a two-kernel hot / cold split: the plan leaves the K most popular columns of every panel OUT of the stream
# (results wrong) -- how fast is the unchanged pair kernel on the cold remainder?
cd $GRAFT_REPO_ROOT
for K in 0 64 128 256; do
echo "== MGGCN_SPMM_DEBUG_DROP_HOT=$K (symmetric stand-in)"
SPMM_AB_SYMMETRIC=1 MGGCN_SPMM_DEBUG_DROP_HOT=$K timeout -k 10 200 python3 profiles/experiments/spmm_ab.py "drop_hot=$K" 2>&1 | grep "drop_hot=\|dropping" | cut -c1-130
done
