for P in 2048 3072 4096; do for S in 48 64 96; do
  MGGCN_SPMM_PANEL_ROWS=$P MGGCN_SPMM_SLICE_MIB=$S python profiles/experiments/spmm_ab.py "panel=$P slice=$S" 2>&1 | tail -1
done; done
MGGCN_SPMM_PANEL_ROWS=4096 MGGCN_SPMM_SLICE_MIB=64 MGGCN_SPMM_PRIO_SHIFT=7 python profiles/experiments/spmm_ab.py "panel=4096 slice=64 shift=7" 2>&1 | tail -1
MGGCN_SPMM_PANEL_ROWS=4096 MGGCN_SPMM_SLICE_MIB=64 MGGCN_SPMM_PRIO_SHIFT=9 python profiles/experiments/spmm_ab.py "panel=4096 slice=64 shift=9" 2>&1 | tail -1
MGGCN_SPMM_PANEL_ROWS=4096 MGGCN_SPMM_SLICE_MIB=64 MGGCN_SPMM_SWEEP_BLOCKS_PER_CU=3 python profiles/experiments/spmm_ab.py "panel=4096 slice=64 blocks/CU=3" 2>&1 | tail -1
