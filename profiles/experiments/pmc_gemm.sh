set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmcgemm
mkdir -p $O
for w in fwd608 bwd608; do
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_LDS -d $O/$w -o sq --output-format csv -- python3 $R/profiles/experiments/one_gemm.py $w 5 > $O/$w.log 2>&1
echo "pass1 $w done"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES -d $O/${w}_b -o sq --output-format csv -- python3 $R/profiles/experiments/one_gemm.py $w 5 > $O/${w}_b.log 2>&1
echo "pass2 $w done"
done
