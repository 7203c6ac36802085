#!/bin/bash
# Runs every variant of tools/probes/rccl_capture_watchdog_probe.py in its own process and tabulates the exit codes.
OUT=gpurun_out/watchdog_probe; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
port=29610
for v in eager_then_capture captured_then_capture captured_held_open captured_eager_capture; do
  for m in global thread_local relaxed; do
    port=$((port+1))
    MASTER_PORT=$port timeout -k 10 120 python tools/probes/rccl_capture_watchdog_probe.py $v $m > $OUT/$v.$m.log 2>&1
    rc=$?
    err=$(grep -o "HIP error: [a-z A-Z]*" $OUT/$v.$m.log | sort -u | tr '\n' ';')
    echo "$v $m rc=$rc $err" | tee -a $OUT/summary.txt
  done
done
