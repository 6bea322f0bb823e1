#!/bin/bash
# The C++ host layer's distributed classes (mg-gcn_amd/host/tests/test_dist.cpp, compiled AS IS) on the stream model:
#   run_host_dist_sim.sh <binary> asan|tsan
# asan: every rank on a device of its own, P = 4 under the three schedules, the push form, P = 3 with an odd shape, the RCCL
#       transport (on the model of hipsim/rccl/rccl.h: its multi-rank branches have never met a second real rank) at P = 4 / 8; then the
#       MUTATION runs -- one kind of event wait ignored at a time (cross-device, join, fork, the host layer's own compute <-> comm
#       edges): each must make cases fail, or the model would prove nothing.
# tsan: the same binary with P enqueue threads under ThreadSanitizer (P = 4 random schedule, P = 8 newest-stream-first, push form).
BIN=$1; KIND=$2
export MGGCN_TEST_DELAY_GEMM=48
fail=0
clean() {   # P policy seed push|rccl [shape...]
  local P=$1 pol=$2 seed=$3 push=$4 tr=p2p; shift 4
  [ "$push" = rccl ] && { tr=rccl; push=0; }
  out=$(MGGCN_COMM_TRANSPORT=$tr HIPSIM_POLICY=$pol HIPSIM_SEED=$seed MGGCN_P2P_PUSH=$push "$BIN" $P "$@" 2>&1); rc=$?
  n=$(echo "$out" | grep -c "TEST PASSED")
  if [ $rc -ne 0 ] || echo "$out" | grep -q "TEST FAILED\|ThreadSanitizer\|AddressSanitizer\|runtime error\|DEADLOCK" || [ "$n" -lt 12 ]; then
    fail=$((fail + 1)); echo "TEST FAILED: host layer on the stream model, P=$P policy=$pol seed=$seed $tr push=$push $*"; echo "$out" | grep -v "TEST PASSED" | tail -15
  else
    echo "TEST PASSED: host layer on the stream model, P=$P policy=$pol seed=$seed $tr push=$push $* ($n cases)"
  fi
}
mutant() {  # class name [transport]
  seen=0
  for pol in 1 0 2; do
    out=$(MGGCN_COMM_TRANSPORT=${3:-p2p} HIPSIM_POLICY=$pol HIPSIM_SEED=7 HIPSIM_DROP_CLASS=$1 "$BIN" 4 2>&1)
    if echo "$out" | grep -q "TEST FAILED"; then seen=1; break; fi
  done
  if [ $seen -eq 1 ]; then echo "TEST PASSED: without its $2 waits the host layer's cases fail (${3:-p2p}: $(echo "$out" | grep -c 'TEST FAILED') of them)"
  else fail=$((fail + 1)); echo "TEST FAILED: without its $2 waits NOT NOTICED (${3:-p2p})"; fi
}
if [ "$KIND" = asan ]; then
  clean 4 0 1 0; clean 4 1 2 0; clean 4 2 4 1; clean 3 1 5 0 393 2 1 7 5
  clean 4 1 7 rccl; clean 8 2 8 rccl
  mutant 1 cross-device; mutant 2 join; mutant 3 fork; mutant 4 "compute <-> comm"; mutant 4 "compute <-> comm" rccl
else
  clean 8 1 2 1; clean 4 0 3 rccl
fi
[ $fail -eq 0 ] && echo "ALL PASSED" || echo "SOME FAILED"
exit $fail
