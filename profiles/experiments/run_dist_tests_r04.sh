set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
B=mg-gcn_amd/bin
for P in ${PS:-1 2 3 4 8}; do
  echo "== test_dist $P"
  if [ $P -gt 1 ]; then export MGGCN_OVERSUBSCRIBE=1; fi
  timeout -k 10 300 $B/test_dist $P > gpurun_out/r04/test_dist_$P.log 2>&1 || { echo "FAILED P=$P"; grep -v "TEST PASSED: dist_gcn" gpurun_out/r04/test_dist_$P.log | tail -30; exit 1; }
  grep -c "TEST PASSED" gpurun_out/r04/test_dist_$P.log; grep "TEST FAILED\|late rank" gpurun_out/r04/test_dist_$P.log || true
done
