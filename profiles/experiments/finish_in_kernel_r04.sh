#!/bin/bash
# in-launch combine of heavy rows against the combine kernel: soak + time, full size and rank 0's share at P = 8
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
{
  timeout -k 10 400 python3 profiles/experiments/finish_in_kernel_r04.py 40 1 &&
  timeout -k 10 300 python3 profiles/experiments/finish_in_kernel_r04.py 60 8
} > gpurun_out/finish_in_kernel_r04.log 2>&1
echo "exit $?" >> gpurun_out/finish_in_kernel_r04.log
tail -30 gpurun_out/finish_in_kernel_r04.log
