set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc41
mkdir -p $O
rocprofv3 -L > $O/counters.txt 2>&1 || true
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD -d $O/sq -o sq --output-format csv -- python3 $R/profiles/experiments/one_spmm.py 41 3 > $O/sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_LDS SQ_WAIT_INST_LDS -d $O/sq2 -o sq2 --output-format csv -- python3 $R/profiles/experiments/one_spmm.py 41 3 > $O/sq2.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum -d $O/tcc -o tcc --output-format csv -- python3 $R/profiles/experiments/one_spmm.py 41 3 > $O/tcc.log 2>&1
# (a fourth pass asked for all TCP_*/TA_* counters at once; rocprofv3 ABORTED with 'error code 38: Request
#  exceeds the capabilities of the hardware to collect' -- gpurun_out/pmc41/tcp.log -- a too-wide --pmc set, not a
#  hang and not a kernel fault.  profiles/experiments/pmc_l1.sh collects them in passes of <= 4 per block.)
ls -R $O | head -40
