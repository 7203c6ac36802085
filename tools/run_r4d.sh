#!/bin/bash
set -o pipefail
OUT=gpurun_out/r4d
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
echo "== memset nodes (rounds 1-3) =="
BDETR_ZERO_MEMSET=1 SYNC_STEPS=8 STEPS=10 step 300 $OUT/ck_memset.log python3 tools/graph_segment_checksums.py
grep -h "^[LFAS] \|eager:" $OUT/ck_memset.log | cut -c1-500
echo "== zero-fill kernel =="
SYNC_STEPS=8 STEPS=10 step 300 $OUT/ck_kernel.log python3 tools/graph_segment_checksums.py
grep -h "^[LFAS] \|eager:" $OUT/ck_kernel.log | cut -c1-500
BDETR_SIDE_STREAM=1 BDETR_GRAPH_SIDE=1 BDETR_DETERMINISTIC=0 SYNC_STEPS=8 STEPS=10 step 300 $OUT/ck_kernel_side_nondet.log python3 tools/graph_segment_checksums.py
grep -h "^[LFAS] \|eager:" $OUT/ck_kernel_side_nondet.log | cut -c1-500
GUARD=1 step 300 $OUT/gd_guard.log python3 tools/graph_debug.py
grep -h "guard tripped\|^graph\|^eager" $OUT/gd_guard.log | cut -c1-500
echo R4D_DONE
