#!/bin/bash
OUT=gpurun_out/r5c
rm -rf $OUT; mkdir -p $OUT
i=0
for f in 2 1 3 2 1 3; do
  i=$((i+1))
  BDETR_WGRAD_WANT_3X3=$f timeout -k 10 240 python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-configs2 --no-fp32-policy --no-batch32 --no-roofline > $OUT/bench_$i.w$f.json 2>> $OUT/bench_err.log || { echo "bench rc=$?"; tail -5 $OUT/bench_err.log; exit 1; }
done
for j in $OUT/bench_*.json; do python3 - "$j" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], d["value"], d["ms_per_step"], d["final_loss"])
PY
done
echo R5C_DONE
