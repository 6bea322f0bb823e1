#!/bin/bash
# soak of the threaded C++ distributed classes on one card: the same binary, many runs, both directions of the peer-copy transport
# (enqueue threads, host sequence counters, event rings: a rare ordering bug shows as a wrong bit, a 120-s abort or a hang)
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
LOG=gpurun_out/dist_soak_r04.log
: > $LOG
fail=0; runs=0
t0=$SECONDS
for i in $(seq 1 12); do
  for cfg in "4|" "8|" "4|MGGCN_P2P_PUSH=1" "8|MGGCN_P2P_PUSH=1" "4|MGGCN_P2P_PEER_STREAMS=1" "3|MGGCN_DIST_CHUNKS=7"; do
    P=${cfg%%|*}; extra=${cfg#*|}
    out=$(env MGGCN_OVERSUBSCRIBE=1 $extra timeout -k 5 120 mg-gcn_amd/bin/test_dist $P 2>&1); rc=$?
    runs=$((runs + 1))
    if [ $rc -ne 0 ] || echo "$out" | grep -q "TEST FAILED"; then
      fail=$((fail + 1)); echo "FAIL run $i P=$P $extra rc=$rc" >> $LOG; echo "$out" | tail -5 >> $LOG
      [ $rc -eq 124 ] && { echo "a run hung: stopping" >> $LOG; break 2; }
    fi
  done
  echo "round $i done: $runs runs, $fail failed, $((SECONDS - t0)) s" >> $LOG
done
echo "SOAK: $runs runs, $fail failed, $((SECONDS - t0)) s" >> $LOG
tail -4 $LOG
