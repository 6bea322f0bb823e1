for T in 1 2 4; do
  MGGCN_SPMM_TASKS_PER_WAVE=$T python profiles/experiments/spmm_ab.py "tasks per wave=$T" 2>&1 | tail -1
done
python profiles/experiments/host_issue_time.py 2>&1 | tail -1
