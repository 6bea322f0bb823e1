# round 3: the narrow (d = 41) kernel's knobs once more after the rotation / interleaved-slice changes, both stand-ins
# bash profiles/experiments/narrow_r03.sh   (GPU box)
cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" python3 profiles/experiments/spmm_ab.py "$*" 2>/dev/null | tail -1; }
run A=default
run MGGCN_SPMM_NARROW_LPE=12
run MGGCN_SPMM_NARROW_LPE=12 MGGCN_SPMM_PANEL_ROWS_NARROW=16384
run MGGCN_SPMM_PANEL_ROWS_NARROW=12288
run MGGCN_SPMM_PANEL_ROWS_NARROW=6144
run MGGCN_SPMM_PRIO_SHIFT_NARROW=0
run MGGCN_SPMM_PRIO_SHIFT_NARROW=2
run MGGCN_SPMM_SWEEP_BLOCKS_PER_CU=3
