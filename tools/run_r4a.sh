#!/bin/bash
# Round 4, GPU call A: deterministic-mode tests, the torch-free replay reproducer, per-segment checksums of the replayed step.
set -o pipefail
OUT=gpurun_out/r4a
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
step() {   # step <seconds> <log> <cmd...>: a step that times out ends the call (no further GPU step after a hang)
  local t=$1 log=$2; shift 2
  timeout -k 10 $t "$@" > $log 2>&1
  local rc=$?
  echo "[$(date +%T)] rc=$rc  $*" | tee -a $OUT/steps.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: stopping" | tee -a $OUT/steps.log; tail -20 $log; exit 1; fi
  return 0
}
step 600 $OUT/pytest_kernels.log python -m pytest tests/test_kernels_gpu.py tests/test_p16_gpu.py -x -q -k "linear_bwd or conv_fwd_bwd or small_activations or splitk"
tail -3 $OUT/pytest_kernels.log
step 600 $OUT/pytest_training.log python -m pytest tests/test_training_gpu.py -x -q
tail -5 $OUT/pytest_training.log
P=tools/probes/graph_replay_repro
step 120 $OUT/repro_default.log $P 16 40 12
step 120 $OUT/repro_default_side.log $P 16 40 12 side
DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 step 120 $OUT/repro_off_side.log $P 16 40 12 side
step 120 $OUT/repro_default_big.log $P 31 60 24 side
grep -h "GRAPH_REPLAY_REPRO\|pass" $OUT/repro_*.log
DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 BDETR_GRAPH_UNSAFE=1 step 300 $OUT/cksum_packets_on.log python3 tools/graph_segment_checksums.py
tail -25 $OUT/cksum_packets_on.log
DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 step 300 $OUT/cksum_packets_off.log python3 tools/graph_segment_checksums.py
tail -6 $OUT/cksum_packets_off.log
echo R4A_DONE
