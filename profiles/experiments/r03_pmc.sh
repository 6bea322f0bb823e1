# round 3: (1) TCP counters of the d = 41 narrow kernel, (2) the two-counter TA / TD passes that were
# scripted in r02 and never run (item 6).  bash profiles/experiments/r03_pmc.sh   (on the GPU box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
pass() {  # pass <outdir> <d> <name> <timeout> counters...
  O=$1; D=$2; name=$3; TO=$4; shift 4
  mkdir -p $O
  timeout -k 10 $TO rocprofv3 --kernel-trace --pmc "$@" -d $O/$name -o $name --output-format csv -- python3 $R/profiles/experiments/one_spmm.py $D 3 > $O/$name.log 2>&1
  echo "$name d=$D rc=$?"
}
O=$R/gpurun_out/pmc_l1_d41
pass $O 41 tcp_a 200 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum
pass $O 41 tcp_b 200 TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
pass $O 41 tcp_c 200 TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum
pass $O 41 sq_a 200 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM
pass $O 41 sq_b 200 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
pass $O 41 grbm 200 GRBM_GUI_ACTIVE GRBM_TA_BUSY
# TA / TD, two counters per pass, 100-s limit: if error 38 again they are dropped for good
O=$R/gpurun_out/pmc_l1_tatd
pass $O 128 ta_a 100 TA_TA_BUSY_sum TA_BUFFER_READ_WAVEFRONTS_sum
pass $O 128 ta_b 100 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
pass $O 128 td_a 100 TD_TD_BUSY_sum TD_TC_STALL_sum
grep -l "error code 38" $O/*.log 2>/dev/null
ls $R/gpurun_out/pmc_l1_d41 $O
