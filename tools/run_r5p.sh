#!/bin/bash
OUT=gpurun_out/r5p
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 900 python3 -m pytest tests/test_kernels_gpu.py tests/test_rowchain_gpu.py tests/test_model_gpu.py tests/test_training_gpu.py -x -q -m gpu > $OUT/tests.log 2>&1; rc=$?; tail -4 $OUT/tests.log | cut -c1-300; echo "tests rc=$rc"
[ $rc -eq 0 ] || exit $rc
for i in 1 2 3; do timeout -k 10 240 python3 bench.py --steps 100 --warmup 10 --graph --no-cpu-baseline --no-configs2 --no-fp32-policy --no-batch32 --no-roofline > $OUT/bench_$i.json 2>> $OUT/bench_err.log || { echo "bench rc=$?"; tail -5 $OUT/bench_err.log; exit 1; }; done
for j in $OUT/bench_*.json; do python3 - "$j" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], d["value"], d["ms_per_step"], d["final_loss"])
PY
done
echo R5P_DONE
