cd $GRAFT_REPO_ROOT
bash profiles/collect.sh r04_final > gpurun_out/collect_r04_final.log 2>&1
echo "collect rc $?"
timeout -k 10 500 python bench.py > gpurun_out/bench_r04_final.json 2> gpurun_out/bench_r04_final.err
echo "bench rc $?"; cut -c1-300 gpurun_out/bench_r04_final.json
