#!/bin/bash
set -o pipefail
OUT=gpurun_out/r4c
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
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 BDETR_GRAPH_UNSAFE=1
SYNC_STEPS=8 STEPS=10 step 300 $OUT/ck_sync8.log python3 tools/graph_segment_checksums.py
grep -h "^[LFAS] \|eager:" $OUT/ck_sync8.log | cut -c1-600
BDETR_SIDE_STREAM=1 BDETR_GRAPH_SIDE=1 SYNC_STEPS=8 STEPS=10 step 300 $OUT/ck_sync8_side.log python3 tools/graph_segment_checksums.py
grep -h "^[LFAS] \|eager:" $OUT/ck_sync8_side.log | cut -c1-600
BDETR_SIDE_STREAM=1 BDETR_GRAPH_SIDE=1 BDETR_DETERMINISTIC=0 SYNC_STEPS=8 STEPS=10 step 300 $OUT/ck_sync8_side_nondet.log python3 tools/graph_segment_checksums.py
grep -h "^[LFAS] \|eager:" $OUT/ck_sync8_side_nondet.log | cut -c1-600
echo R4C_DONE
