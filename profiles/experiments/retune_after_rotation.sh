# panel / slice re-tuning with the priority rotation on (run on the GPU box)
python profiles/experiments/spmm_ab.py "defaults" 2>&1 | tail -1
for P in 4096 6144 8192 12288; do for S in 32 64 128; do
  MGGCN_SPMM_PANEL_ROWS=$P MGGCN_SPMM_SLICE_MIB=$S python profiles/experiments/spmm_ab.py "panel=$P slice=$S" 2>&1 | tail -1
done; done
for P in 8192 16384 32768; do
  MGGCN_SPMM_PANEL_ROWS_NARROW=$P python profiles/experiments/spmm_ab.py "narrow panel=$P" 2>&1 | tail -1
done
