#!/bin/bash
OUT=gpurun_out/r4s
rm -rf $OUT; mkdir -p $OUT
BDETR_PROF_DUMP=$OUT/launches.csv timeout -k 10 300 python3 bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-configs2 --no-graph > $OUT/bench.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
python3 tools/launch_roofline.py $OUT/launches.csv 5 > $OUT/launch_roofline.txt; head -60 $OUT/launch_roofline.txt
echo R4S_DONE
