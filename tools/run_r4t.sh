#!/bin/bash
OUT=gpurun_out/r4t
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 600 python3 -m pytest "tests/test_training_gpu.py::test_gradient_policy_of_its_own_runs_the_backward_under_it" "tests/test_fullsize_gpu.py::test_config2_batch2_matches_oracle" -x -q -s -m gpu > $OUT/tests.log 2>&1; rc=$?; grep -a "gradient error\|passed\|failed\|Error\|assert" $OUT/tests.log | cut -c1-600 | tail -12; echo "tests rc=$rc"
[ $rc -eq 0 ] || { tail -30 $OUT/tests.log; }
timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 --no-configs2 > $OUT/bench.json 2> $OUT/bench.err || { echo "bench rc=$?"; tail -5 $OUT/bench.err; exit 1; }
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r4t/bench.json').read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["value_fp32_policy"], d["value_fp32_grade"])
PY
echo R4T_DONE
