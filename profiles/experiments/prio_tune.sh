# rotation period sweep (run on the GPU box)
MGGCN_SPMM_PRIO_ROTATE=0 python profiles/experiments/spmm_ab.py "no rotation" 2>&1 | tail -1
for S in 5 6 7 8 9 10; do
  MGGCN_SPMM_PRIO_SHIFT=$S python profiles/experiments/spmm_ab.py "wide shift=$S narrow=0" 2>&1 | tail -1
done
for S in 1 2 3; do
  MGGCN_SPMM_PRIO_SHIFT_NARROW=$S python profiles/experiments/spmm_ab.py "wide shift=8 narrow=$S" 2>&1 | tail -1
done
