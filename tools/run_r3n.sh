#!/bin/bash
set -o pipefail
OUT=gpurun_out/r3n
mkdir -p $OUT
export TMPDIR=/tmp
BDETR_SIDE_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 bench.py --steps 5 --warmup 2 --no-graph --no-cpu-baseline --no-roofline --no-batch32 --no-fp32-policy > $OUT/prof.log 2>&1 || { tail -20 $OUT/prof.log; exit 1; }
find $OUT/prof -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_serial.csv \;
find $OUT/prof -name "*kernel_trace.csv" -delete
python tools/kstats.py $OUT/kernel_stats_serial.csv 7 0.3
echo R3N_DONE
