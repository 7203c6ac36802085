#!/bin/bash
set -o pipefail
OUT=gpurun_out/r3f
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_training_gpu.py tests/test_model_gpu.py tests/test_precision_gpu.py -q 2>&1 | tee $OUT/tests.log | tail -25
rc=${PIPESTATUS[0]}
if [ $rc -ge 124 ]; then echo "pytest killed"; exit 1; fi
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-batch32 --no-fp32-policy > $OUT/bench.json 2> $OUT/bench.err || { tail -30 $OUT/bench.err; exit 1; }
BDETR_ATTN_SPLIT=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-batch32 --no-fp32-policy --no-roofline > $OUT/bench_noattn.json 2> $OUT/bench_noattn.err || { tail -30 $OUT/bench_noattn.err; exit 1; }
python - <<'PY'
import json
for f in ('bench','bench_noattn'):
    d=json.loads(open(f'gpurun_out/r3f/{f}.json').read().strip().splitlines()[-1])
    print(f,'images/s',d['value'],'ms',d['ms_per_step'], d['final_loss'], d['config']['env_overrides'])
PY
BDETR_SIDE_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-batch32 --no-fp32-policy > $OUT/prof.log 2>&1 || { tail -20 $OUT/prof.log; exit 1; }
find $OUT/prof -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_serial.csv \;
find $OUT/prof -name "*kernel_trace.csv" -delete
python tools/kstats.py $OUT/kernel_stats_serial.csv 7 0.25
timeout -k 10 200 python tools/dense_probe.py 2>&1 | tee $OUT/dense_probe.log
echo R3F_DONE tests_rc=$rc
