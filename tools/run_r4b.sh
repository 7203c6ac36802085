#!/bin/bash
# Round 4, GPU call B: does the round-3 replay failure still reproduce, and under which conditions?
set -o pipefail
OUT=gpurun_out/r4b
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
step() {
  local t=$1 log=$2; shift 2
  timeout -k 10 $t "$@" > $log 2>&1
  local rc=$?
  echo "[$(date +%T)] rc=$rc  $*" | tee -a $OUT/steps.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: stopping" | tee -a $OUT/steps.log; tail -20 $log; exit 1; fi
  return 0
}
step 600 $OUT/pytest_kernels.log python -m pytest tests/test_kernels_gpu.py tests/test_p16_gpu.py -x -q
tail -3 $OUT/pytest_kernels.log
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 BDETR_GRAPH_UNSAFE=1
# round 3's reproducer as it was (atomics, side stream, guard on)
GUARD=1 step 300 $OUT/gd_guard.log python3 tools/graph_debug.py
tail -12 $OUT/gd_guard.log
GUARD=1 BDETR_DETERMINISTIC=1 step 300 $OUT/gd_guard_det.log python3 tools/graph_debug.py
tail -12 $OUT/gd_guard_det.log
GUARD=0 step 300 $OUT/gd_noguard.log python3 tools/graph_debug.py
tail -6 $OUT/gd_noguard.log
# checksum tool, non-deterministic (only the non-finite counts are meaningful), more steps
BDETR_DETERMINISTIC=0 STEPS=24 step 300 $OUT/ck_nondet.log python3 tools/graph_segment_checksums.py
tail -4 $OUT/ck_nondet.log
STEPS=24 step 300 $OUT/ck_det24.log python3 tools/graph_segment_checksums.py
tail -4 $OUT/ck_det24.log
echo R4B_DONE
