#!/bin/bash
OUT=gpurun_out/r3o
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_p16_gpu.py tests/test_training_gpu.py tests/test_kernels_gpu.py -x -q 2>&1 | tail -4
for i in 1 2; do
timeout -k 10 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --no-batch32 --no-fp32-policy > $OUT/bench_$i.json 2> $OUT/bench_$i.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3o/bench_*.json')):
    try:
        o=json.loads(open(f).read().strip().split('\n')[-1]); print(f, o['value'], o['ms_per_step'], o['config']['step_launch'], o['final_loss'])
    except Exception as e: print(f, 'ERR', e)
PY
echo R3O_DONE
