cd $GRAFT_REPO_ROOT
for r in 0 16 0 16; do echo "MGGCN_SPMM_RESERVED_CUS=$r"; MGGCN_SPMM_RESERVED_CUS=$r python3 profiles/experiments/rank_epoch_r04.py 2>/dev/null | grep "^P="; done
