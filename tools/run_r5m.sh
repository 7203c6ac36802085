#!/bin/bash
OUT=gpurun_out/r5m
rm -rf $OUT; mkdir -p $OUT
B="--steps 60 --warmup 5 --no-cpu-baseline --no-configs2 --no-fp32-policy --no-batch32 --no-roofline"
for m in "" "--graph" "--no-graph" ""; do
  timeout -k 10 240 python3 bench.py $B $m > $OUT/bench_$RANDOM.json 2>> $OUT/bench_err.log || { echo "bench rc=$?"; tail -5 $OUT/bench_err.log; exit 1; }
done
for j in $OUT/bench_*.json; do python3 - "$j" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["config"]["step_launch"], d["config"]["launch_probe"])
PY
done
grep "launch probe" $OUT/bench_err.log
# data-parallel code path on a one-rank communicator, default flags (probe) and the forced-graph test
MASTER_ADDR=127.0.0.1 MASTER_PORT=29577 WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 BDETR_DP_FORCE=1 timeout -k 10 300 python3 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-batch32 --no-fp32-policy > $OUT/dp.json 2> $OUT/dp.err; echo "dp rc=$?"; grep "launch probe" $OUT/dp.err; python3 -c "
import json; d=json.loads(open('$OUT/dp.json').read().strip().splitlines()[-1]); print(d['value'], d['config']['step_launch'], d['config']['launch_probe'], d['config']['distributed']['world_size'])"
timeout -k 10 600 python3 -m pytest tests/test_dp_gpu.py -x -q -m gpu > $OUT/tests.log 2>&1; rc=$?; tail -3 $OUT/tests.log; echo "tests rc=$rc"
echo R5M_DONE
