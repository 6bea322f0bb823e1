cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout -k 10 1000 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_host_cpp.py tests/test_gpu_bench.py tests/test_gpu_gcn.py -x -q -m gpu > gpurun_out/r04/subset.log 2>&1
echo "rc $?"; tail -15 gpurun_out/r04/subset.log
