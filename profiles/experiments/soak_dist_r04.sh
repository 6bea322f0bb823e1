# bounded soak of the threaded peer-copy protocol (host sequence counters, event ring): the same binary, several times, P = 3 / 4 / 8
cd $GRAFT_REPO_ROOT
export MGGCN_OVERSUBSCRIBE=1
for it in 1 2 3 4 5; do
  for P in 3 4 8; do
    timeout -k 10 200 mg-gcn_amd/bin/test_dist $P > /tmp/soak.log 2>&1
    rc=$?
    echo "iteration $it P=$P rc=$rc passed=$(grep -c 'TEST PASSED' /tmp/soak.log) failed=$(grep -c 'TEST FAILED\|FAILURE' /tmp/soak.log)"
    if [ $rc -ne 0 ]; then grep -v "TEST PASSED" /tmp/soak.log | tail -20; exit 1; fi
  done
done
# many exchanges in a row with no release in between (the event ring wraps, forced releases): MGGCN_DIST_CHUNKS=40 pieces per SpMM
MGGCN_DIST_CHUNKS=40 timeout -k 10 200 mg-gcn_amd/bin/test_dist 4 1536 42 24 6 32 16 > /tmp/soak.log 2>&1; echo "chunks=40 P=4 rc=$? passed=$(grep -c 'TEST PASSED' /tmp/soak.log)"
MGGCN_DIST_CHUNKS=70 timeout -k 10 200 mg-gcn_amd/bin/test_dist 3 1536 42 24 6 32 16 > /tmp/soak.log 2>&1; echo "chunks=70 P=3 rc=$? passed=$(grep -c 'TEST PASSED' /tmp/soak.log)"
