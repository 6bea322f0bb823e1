# usage: bash profiles/experiments/pmc_spmm.sh <d> <tag>   (run on the GPU box through gpurun)
set -e
D=$1; TAG=$2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_$TAG
mkdir -p $O
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD -d $O/sq -o sq --output-format csv -- python3 $R/profiles/experiments/one_spmm.py $D 3 > $O/sq.log 2>&1
echo "sq done"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum -d $O/tcc -o tcc --output-format csv -- python3 $R/profiles/experiments/one_spmm.py $D 3 > $O/tcc.log 2>&1
echo "tcc done"
