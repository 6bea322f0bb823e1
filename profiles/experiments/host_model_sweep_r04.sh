#!/bin/bash
BIN=/root/repo/tests/native/_build/host_dist_sim_asan
export MGGCN_TEST_DELAY_GEMM=48 MGGCN_COMM_TRANSPORT=${SWEEP_TRANSPORT:-p2p} ASAN_OPTIONS=detect_leaks=0
run() {
  P=$1 pol=$2 seed=$3 push=$4 chunks=$5
  out=$(HIPSIM_POLICY=$pol HIPSIM_SEED=$seed MGGCN_P2P_PUSH=$push MGGCN_DIST_CHUNKS=$chunks timeout 300 $BIN $P 2>&1); rc=$?
  if [ $rc -ne 0 ] || echo "$out" | grep -q "TEST FAILED\|AddressSanitizer\|runtime error\|DEADLOCK"; then
    echo "FAIL P=$P pol=$pol seed=$seed push=$push chunks=$chunks rc=$rc"; echo "$out" | grep -v "TEST PASSED" | tail -8
  else echo "ok P=$P pol=$pol seed=$seed push=$push chunks=$chunks"; fi
}
export -f run; export BIN
for P in 2 3 4 6 8; do for pol in 0 1 2; do for seed in 11 12 13; do for push in 0 1; do for chunks in 1 3 40; do
  [ $pol -ne 0 ] && [ $seed -ne 11 ] && continue
  echo "$P $pol $seed $push $chunks"
done; done; done; done; done | xargs -P 6 -L 1 bash -c 'run $0 $1 $2 $3 $4' > /tmp/sweep_model.log 2>&1
echo "done: $(grep -c '^ok' /tmp/sweep_model.log) ok, $(grep -c '^FAIL' /tmp/sweep_model.log) failed" >> /tmp/sweep_model.log
