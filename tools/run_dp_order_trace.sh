#!/bin/bash
# kernel-trace statistics of the data-parallel step (one-rank RCCL) with the process group created BEFORE and AFTER the model (tools/dp_gc_probe.py)
OUT=gpurun_out/dp_order
rm -rf $OUT; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0 BDETR_DP_FORCE=1 PROBE_SHORT=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
PROBE_INIT_FIRST=1 timeout -k 10 280 rocprofv3 --kernel-trace --stats -d $OUT/first -o first -- python3 tools/dp_gc_probe.py > $OUT/first.log 2>&1 || { tail -5 $OUT/first.log; exit 1; }
PROBE_INIT_FIRST=0 timeout -k 10 280 rocprofv3 --kernel-trace --stats -d $OUT/after -o after -- python3 tools/dp_gc_probe.py > $OUT/after.log 2>&1 || { tail -5 $OUT/after.log; exit 1; }
grep "collector on" $OUT/first.log $OUT/after.log | cut -c1-120
find $OUT -name "*kernel_stats.csv" | head
# keep the stats, drop the traces (large)
find $OUT -name "*kernel_trace.csv" -size +60M -delete
echo TRACE_DONE
