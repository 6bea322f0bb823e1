cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest tests -x -q -m gpu --durations=25 > gpurun_out/r04/full_final.log 2>&1
echo "tests rc $?"; tail -4 gpurun_out/r04/full_final.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -k 10 400 python bench.py > gpurun_out/bench_r04_final.json 2> gpurun_out/bench_r04_final.err
echo "bench rc $?"; cut -c1-200 gpurun_out/bench_r04_final.json
