#!/bin/bash
OUT=gpurun_out/r4i
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 300 python3 tools/rowchain_bench.py > $OUT/rc_bench.log 2>&1; echo "rc=$?"; grep "M=" $OUT/rc_bench.log
timeout -k 10 600 python -m pytest tests/test_rowchain_gpu.py tests/test_kernels_gpu.py -q -k "rowchain or fused_block or grouped" > $OUT/pytest_a.log 2>&1; echo "rc=$?"; tail -3 $OUT/pytest_a.log
timeout -k 10 900 python -m pytest tests/test_model_gpu.py tests/test_training_gpu.py tests/test_fullsize_gpu.py tests/test_precision_gpu.py tests/test_dp_gpu.py tests/test_edge_cases_gpu.py -x -q > $OUT/pytest_model.log 2>&1; echo "rc=$?"; tail -15 $OUT/pytest_model.log
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-configs2 --no-batch32 --no-fp32-policy > $OUT/bench.json 2> $OUT/bench.err; echo "rc=$?"; tail -2 $OUT/bench.err; python - <<'PY'
import json
o=json.loads(open('gpurun_out/r4i/bench.json').read().strip().split('\n')[-1])
print(o['value'], o['ms_per_step'], o['config']['step_launch'], o['final_loss'], o['roofline']['frac'], o['roofline']['launches_per_step'])
for k,v in o['roofline']['by_class'].items(): print(k, v)
PY
B="--no-cpu-baseline --no-roofline --no-batch32 --no-fp32-policy --no-configs2"
BDETR_SIDE_STREAM=0 timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/serial -- python3 bench.py --steps 5 --warmup 2 --no-graph $B > $OUT/serial.log 2>&1; echo "rc=$?"
find $OUT/serial -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_serial.csv \;
find $OUT -name "*kernel_trace.csv" -delete
python tools/kstats.py $OUT/kernel_stats_serial.csv 7 0.2 | head -50
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r4i/kernel_stats_serial.csv')))
print('launches per step', sum(int(r['Calls']) for r in rows)/7)
PY
echo R4I_DONE
