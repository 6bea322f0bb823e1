# usage: bash profiles/experiments/pmc_l1.sh <d> <tag>   (run on the GPU box through gpurun)
# Vector-L1 (TCP) / texture-address (TA) / texture-data (TD) counters of ONE SpMM shape, in passes
# of at most 4 counters per hardware block.  Round 1 asked for all of them in one pass and rocprofv3
# aborted with "error code 38: Request exceeds the capabilities of the hardware to collect"
# (gpurun_out/pmc41/tcp.log) -- a too-wide --pmc set, not a kernel fault.  Every pass is its own
# process under its own timeout; a pass that the profiler rejects is reported and skipped.
D=$1; TAG=$2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_l1_$TAG
mkdir -p $O
pass() {
  name=$1; shift
  timeout -k 10 ${TO:-240} rocprofv3 --kernel-trace --pmc "$@" -d $O/$name -o $name --output-format csv -- python3 $R/profiles/experiments/one_spmm.py $D 3 > $O/$name.log 2>&1
  echo "$name rc=$?"
}
pass tcp_a TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum
pass tcp_b TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
pass tcp_c TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum
# TA_* / TD_* with FOUR counters per pass (r02, first try): error 38 again ("exceeds the capabilities of the
# hardware") -- those blocks take fewer counters per pass than TCP -- and the aborted rocprofv3 then lingered until
# this script's timeout killed it (rc 124; no kernel ever ran, no fault).  Two per pass, short timeout:
TO=100
pass ta_a TA_TA_BUSY_sum TA_BUFFER_READ_WAVEFRONTS_sum
pass ta_b TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
pass td_a TD_TD_BUSY_sum TD_TC_STALL_sum
TO=240
pass sq_a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM
pass grbm GRBM_GUI_ACTIVE GRBM_TA_BUSY
ls $O
