cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout -k 10 1100 python -m pytest tests -x -q -m gpu --durations=30 --deselect tests/test_abi_host.py::test_traffic_counters_belong_to_the_kernels_in_the_tree > gpurun_out/r04/full_durations.log 2>&1
echo "rc $?"; grep -A 34 "slowest" gpurun_out/r04/full_durations.log | cut -c1-150
