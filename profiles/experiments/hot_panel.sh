# usage (GPU box): bash profiles/experiments/hot_panel.sh
# the panel_rows most popular columns as a panel of their own, swept first (MGGCN_SPMM_HOT_PANEL=1)
run() { env "$@" python profiles/experiments/spmm_ab.py "$*" 2>&1 | tail -1; }
run MGGCN_SPMM_HOT_PANEL=0
run MGGCN_SPMM_HOT_PANEL=1
run MGGCN_SPMM_HOT_PANEL=1 MGGCN_SPMM_PANEL_ROWS=8192
run MGGCN_SPMM_HOT_PANEL=1 MGGCN_SPMM_PANEL_ROWS=2048
run MGGCN_SPMM_HOT_PANEL=0
