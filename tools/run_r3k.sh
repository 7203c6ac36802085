#!/bin/bash
set -o pipefail
OUT=gpurun_out/r3k
mkdir -p $OUT
export TMPDIR=/tmp
bash tools/run_pmc3x3.sh > $OUT/pmc.log 2>&1 || { tail -20 $OUT/pmc.log; exit 1; }
tail -3 $OUT/pmc.log
BDETR_SIDE_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 bench.py --steps 5 --warmup 2 --no-graph --no-cpu-baseline --no-roofline --no-batch32 --no-fp32-policy > $OUT/prof.log 2>&1 || { tail -20 $OUT/prof.log; exit 1; }
find $OUT/prof -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_serial.csv \;
find $OUT/prof -name "*kernel_trace.csv" -delete
python tools/kstats.py $OUT/kernel_stats_serial.csv 7 0.2
echo R3K_DONE
