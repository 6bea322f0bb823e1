python profiles/experiments/spmm_ab.py "defaults" 2>&1 | tail -1
MGGCN_SPMM_NARROW_LPE=12 python profiles/experiments/spmm_ab.py "narrow lpe=12" 2>&1 | tail -1
for P in 4096 6144 12288; do
  MGGCN_SPMM_PANEL_ROWS_NARROW=$P python profiles/experiments/spmm_ab.py "narrow panel=$P" 2>&1 | tail -1
done
MGGCN_SPMM_NARROW_LPE=12 MGGCN_SPMM_PANEL_ROWS_NARROW=6144 python profiles/experiments/spmm_ab.py "narrow lpe=12 panel=6144" 2>&1 | tail -1
