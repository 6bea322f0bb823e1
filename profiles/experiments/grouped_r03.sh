cd $GRAFT_REPO_ROOT
for B in 4 3 2; do for P in 4096 2048 8192; do
echo "== blocks_per_cu=$B panel_rows=$P"; MGGCN_SPMM_SWEEP_BLOCKS_PER_CU=$B MGGCN_SPMM_PANEL_ROWS=$P python profiles/experiments/community_r03.py 2>&1 | grep "community.*d=128" | cut -c1-60
done; done
