# FAST form of the pair kernel (power-of-two pitch: no mask, OR for ADD, v_bfi) against the general form: parity, then time
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python3 -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "sweep or random or spmm" 2>&1 | tail -n 3
for SYM in 0 1; do
for V in 0 1; do
echo "== symmetric=$SYM MGGCN_SPMM_FAST_PAIRS=$V"
SPMM_AB_SYMMETRIC=$SYM MGGCN_SPMM_FAST_PAIRS=$V timeout -k 10 200 python3 profiles/experiments/spmm_ab.py "fast=$V" 2>&1 | grep "fast=" | cut -c1-200
done
done
