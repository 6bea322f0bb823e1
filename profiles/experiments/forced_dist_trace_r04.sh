# where do the +16 % of SpMM time under an active exchange come from?
# kernel trace (start/end of every launch) of the N > 1 code path with one RCCL rank, and of the plain N = 1 path
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_forced_dist
mkdir -p $O
B="python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras"
export MGGCN_BENCH_FORCE_DIST=1
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/dist -o dist --output-format csv -- $B > $O/dist_bench.json 2> $O/dist.err
echo "dist trace done"
unset MGGCN_BENCH_FORCE_DIST
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/plain -o plain --output-format csv -- $B > $O/plain_bench.json 2> $O/plain.err
echo "plain trace done"
ls -la $O/*/
