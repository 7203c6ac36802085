#!/bin/bash
# stem tail fusion: parity, model-level tests, one-box A/B
OUT=gpurun_out/r4q
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 500 python3 -m pytest tests/test_stem_gpu.py tests/test_model_gpu.py -x -q -m gpu > $OUT/tests.log 2>&1; rc=$?; tail -15 $OUT/tests.log; echo "tests rc=$rc"
[ $rc -eq 0 ] || exit $rc
for f in 0 1 0 1; do
  BDETR_STEM_FUSE=$f timeout -k 10 240 python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-configs2 > $OUT/bench_fuse$f.$RANDOM.json 2>> $OUT/bench_err.log || { echo "bench rc=$?"; tail -5 $OUT/bench_err.log; exit 1; }
done
for j in $OUT/bench_fuse*.json; do python3 - "$j" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], d["value"], d["ms_per_step"])
PY
done
echo R4Q_DONE
