cd $GRAFT_REPO_ROOT
python3 profiles/experiments/gemm_small_m_r04.py 2>/dev/null
MGGCN_GEMM_WIDE_TILES=1 python3 profiles/experiments/gemm_small_m_r04.py 2>/dev/null
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -k gemm 2>&1 | tail -2
python3 profiles/experiments/rank_epoch_r04.py 2>/dev/null | grep "^P="
