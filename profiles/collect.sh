# Collects the rocprofv3 evidence kept under profiles/ (run on the GPU box through gpurun):
#   bash profiles/collect.sh <tag>
# 1. --kernel-trace --stats of the default bench command (per-kernel time)
# 2.-4. separate --pmc passes (FETCH_SIZE / WRITE_SIZE / L2 hit+miss), as the MI355X guide prescribes
# Afterwards, in the repo:  python profiles/summarize.py <tag> gpurun_out/prof_<tag>/stats/..._kernel_stats.csv \
#     --fetch ... --write ... --l2 ... --main-kernel spmm_sweep_pair_kernel --launches-per-unit N
set -e
TAG=$1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
BENCH="python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras ${BENCH_ARGS:-}"   # BENCH_ARGS=--symmetric: the symmetric stand-in
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/stats -o stats --output-format csv -- $BENCH > $O/bench_under_rocprof.json 2> $O/stats.err
echo "stats done"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o fetch --output-format csv -- $BENCH > /dev/null 2> $O/fetch.err
echo "fetch done"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o write --output-format csv -- $BENCH > /dev/null 2> $O/write.err
echo "write done"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d $O/l2 -o l2 --output-format csv -- $BENCH > /dev/null 2> $O/l2.err
echo "l2 done"
ls $O/*/ | head -40
