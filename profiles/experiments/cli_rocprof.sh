# round 3: the reference's interface under the profiler -- which kernels does `mg_gcn train <reddit> 3 128 128 128` run?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_cli
mkdir -p $O /tmp/cli_p
python3 - <<PY
import os, sys
sys.path.insert(0, "$R")
import __graft_entry__ as ge
pkg = ge.load_package()
(ip, ix, dv), X, Y = pkg.datasets.synth_reddit_like(1.0, seed=1)
pkg.datasets.write_dataset("/tmp/cli_p/permuted/reddit", ip, ix, dv, X, Y)
PY
cd /tmp/cli_p
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats -o stats --output-format csv -- $R/mg-gcn_amd/bin/mg_gcn -E 8 train /tmp/cli_p/permuted/reddit 3 128 128 128 > $O/cli.out 2> $O/cli.err
echo rc=$?
grep -E "^[0-9]+ " $O/cli.err | tail -3
head -12 $O/stats/stats_kernel_stats.csv | cut -c1-150
