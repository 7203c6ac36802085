#!/bin/bash
OUT=gpurun_out/r3q
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_p16_gpu.py tests/test_training_gpu.py tests/test_model_gpu.py tests/test_fullsize_gpu.py -x -q 2>&1 | tail -4
for i in 1 2; do
BDETR_WGRAD_XF16=0 timeout -k 10 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --no-batch32 --no-fp32-policy > $OUT/bench_off_$i.json 2> $OUT/bench_off_$i.err
timeout -k 10 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --no-batch32 --no-fp32-policy > $OUT/bench_on_$i.json 2> $OUT/bench_on_$i.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3q/bench_*.json')):
    try:
        o=json.loads(open(f).read().strip().split('\n')[-1]); print(f, o['value'], o['ms_per_step'], o['config']['step_launch'], o['final_loss'], o['config']['env_overrides'])
    except Exception as e: print(f, 'ERR', e)
PY
timeout -k 10 300 python tools/p16_bench.py 16 wgrad 2>&1 | tail -1
echo R3Q_DONE
