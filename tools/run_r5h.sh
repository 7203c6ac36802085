#!/bin/bash
OUT=gpurun_out/r5h
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 1100 python3 -m pytest tests/test_precision_gpu.py -x -q -m gpu -s -k "bf16x6 or three_term or default_policy" > $OUT/tests.log 2>&1; rc=$?; grep -a "fwd\|passed\|failed\|Error\|assert" $OUT/tests.log | cut -c1-700 | tail -12; echo "tests rc=$rc"
[ $rc -eq 0 ] || { tail -30 $OUT/tests.log | cut -c1-300; exit $rc; }
timeout -k 10 600 python3 -m pytest "tests/test_fullsize_gpu.py::test_config2_batch2_matches_oracle" -x -q -s -m gpu > $OUT/tests2.log 2>&1; rc=$?; grep -a "gradient error\|bf16x6\|passed\|failed\|Error\|assert" $OUT/tests2.log | cut -c1-600 | tail -8; echo "tests2 rc=$rc"
timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 --no-configs2 --no-cpu-baseline --no-batch32 > $OUT/bench.json 2> $OUT/bench.err || { echo "bench rc=$?"; tail -5 $OUT/bench.err; exit 1; }
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r5h/bench.json').read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["value_fp32_policy"]["value"], {k: (v if k!="with_exact_fp32_backward" else v["value"]) for k,v in d["value_fp32_grade"].items() if k in ("value","ms_per_step","with_exact_fp32_backward")})
PY
echo R5H_DONE
