# A/B of the priority rotation in spmm_sweep_pair_kernel (run on the GPU box): wave-time spread by hardware slot + SpMM times
for R in 0 1; do
  echo "=== MGGCN_SPMM_PRIO_ROTATE=$R"
  MGGCN_SPMM_PRIO_ROTATE=$R python profiles/experiments/wave_spread.py 2>&1 | grep -A2 "launch 1:" | grep -v "SIMD\|^--" | head -8 | cut -c1-330
  MGGCN_SPMM_PRIO_ROTATE=$R python profiles/experiments/spmm_ab.py "prio_rotate=$R" 2>&1 | tail -1
done
