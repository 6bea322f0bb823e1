# round 3: the panel / slice knobs once more, on the SYMMETRIC stand-in (they were tuned on the asymmetric one in r02)
cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" SPMM_AB_SYMMETRIC=1 python3 profiles/experiments/spmm_ab.py "$*" 2>/dev/null | tail -1; }
run A=default
run MGGCN_SPMM_PANEL_ROWS=3072
run MGGCN_SPMM_PANEL_ROWS=6144
run MGGCN_SPMM_PANEL_ROWS=8192
run MGGCN_SPMM_SLICE_MIB=32
run MGGCN_SPMM_SLICE_MIB=128
run MGGCN_SPMM_PRIO_SHIFT=7
run MGGCN_SPMM_PRIO_SHIFT=9
run MGGCN_SPMM_SWEEP_SPLIT=2048
run MGGCN_SPMM_SWEEP_SPLIT=8192
