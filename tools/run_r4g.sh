#!/bin/bash
OUT=gpurun_out/r4g
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
B="--no-cpu-baseline --no-roofline --no-batch32 --no-fp32-policy --no-configs2"
BDETR_SIDE_STREAM=0 timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/serial -- python3 bench.py --steps 5 --warmup 2 --no-graph $B > $OUT/serial.log 2>&1; echo "rc=$?"
find $OUT/serial -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_serial.csv \;
find $OUT -name "*kernel_trace.csv" -delete
python tools/kstats.py $OUT/kernel_stats_serial.csv 7 0.25 | head -60
echo R4G_DONE
