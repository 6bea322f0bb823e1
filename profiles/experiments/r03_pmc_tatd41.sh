# TA / TD counters of the d = 41 narrow kernel (two per pass), to go with profiles/r03_l1_pmc_d41_summary.md
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_l1_tatd41
mkdir -p $O
pass() { name=$1; shift; timeout -k 10 100 rocprofv3 --kernel-trace --pmc "$@" -d $O/$name -o $name --output-format csv -- python3 $R/profiles/experiments/one_spmm.py 41 3 > $O/$name.log 2>&1; echo "$name rc=$?"; }
pass ta_a TA_TA_BUSY_sum TA_BUFFER_READ_WAVEFRONTS_sum
pass ta_b TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
pass td_a TD_TD_BUSY_sum TD_TC_STALL_sum
