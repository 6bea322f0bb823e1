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
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc "$@" -d $O/$name -o $name --output-format csv -- python3 $R/profiles/experiments/one_spmm.py $D 3 > $O/$name.log 2>&1
  echo "$name rc=$?"
}
pass tcp_a TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum
pass tcp_b TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
pass tcp_c TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum
# TA_* / TD_* passes (TA_TA_BUSY_sum TA_BUFFER_READ_WAVEFRONTS_sum ..., TD_TD_BUSY_sum TD_TC_STALL_sum ...): r02 ran them
# once -- each sat until this script's own 240-s timeout killed rocprofv3 (rc 124, no output, no kernel fault):
# the profiler does not return for those two blocks on this image.  Not collected; do not re-enable.
pass sq_a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM
pass grbm GRBM_GUI_ACTIVE GRBM_TA_BUSY
ls $O
