#!/bin/bash
OUT=gpurun_out/r4x
rm -rf $OUT; mkdir -p $OUT
B="--steps 100 --warmup 10 --no-cpu-baseline --no-configs2 --no-fp32-policy --no-batch32 --no-roofline"
for seg in 10 3 5 7 14 10; do
  BDETR_GRAPH_SEG=$seg timeout -k 10 240 python3 bench.py $B > $OUT/bench_seg$seg.$RANDOM.json 2>> $OUT/bench_err.log || { echo "bench rc=$?"; tail -5 $OUT/bench_err.log; exit 1; }
done
for i in 1 2; do timeout -k 10 240 python3 bench.py $B --no-graph > $OUT/bench_eager.$i.json 2>> $OUT/bench_err.log || { echo "bench rc=$?"; tail -5 $OUT/bench_err.log; exit 1; }; done
for j in $OUT/bench_*.json; do python3 - "$j" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], d["value"], d["ms_per_step"], d["config"]["step_launch"])
PY
done
echo R4X_DONE
