#!/bin/bash
# graph replay with BDETR_GRAPH_SEG (tape nodes per main segment) swept next to eager steps, alternating, on one box
OUT=gpurun_out/segsweep
rm -rf $OUT; mkdir -p $OUT
B="--steps 100 --warmup 10 --no-cpu-baseline --no-configs2 --no-fp32-policy --no-batch32 --no-roofline"
for rep in 1 2; do
for seg in 10 1 2 3 5; do
  BDETR_GRAPH_SEG=$seg timeout -k 10 240 python3 bench.py $B --graph > $OUT/bench_seg$seg.$rep.json 2>> $OUT/bench_err.log || { echo "bench rc=$?"; tail -5 $OUT/bench_err.log; exit 1; }
done
timeout -k 10 240 python3 bench.py $B --no-graph > $OUT/bench_eager.$rep.json 2>> $OUT/bench_err.log || { echo "bench rc=$?"; tail -5 $OUT/bench_err.log; exit 1; }
done
for j in $OUT/bench_*.json; do python3 - "$j" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], d["value"], d["ms_per_step"], d["config"]["step_launch"])
PY
done
echo SEGSWEEP_DONE
