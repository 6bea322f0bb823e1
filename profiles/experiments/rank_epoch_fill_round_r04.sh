cd $GRAFT_REPO_ROOT
for r in default 0 default 0; do echo "MGGCN_SPMM_RESERVED_CUS=$r"; if [ $r = default ]; then python3 profiles/experiments/rank_epoch_r04.py 2>/dev/null | grep "^P="; else MGGCN_SPMM_RESERVED_CUS=$r python3 profiles/experiments/rank_epoch_r04.py 2>/dev/null | grep "^P="; fi; done
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz_model.py -x -q 2>&1 | tail -2
