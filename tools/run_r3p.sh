#!/bin/bash
OUT=gpurun_out/r3p
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_p16_gpu.py tests/test_kernels_gpu.py tests/test_training_gpu.py tests/test_model_gpu.py -x -q 2>&1 | tail -5
python tools/dgrad_epi_probe.py 2>&1 | grep -v amdgpu.ids | tee $OUT/dgrad_epi.log
timeout -k 10 300 python tools/p16_bench.py 16 p16 > $OUT/p16.log 2>&1; tail -1 $OUT/p16.log
for i in 1 2; do
timeout -k 10 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --no-batch32 --no-fp32-policy > $OUT/bench_$i.json 2> $OUT/bench_$i.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3p/bench_*.json')):
    try:
        o=json.loads(open(f).read().strip().split('\n')[-1]); print(f, o['value'], o['ms_per_step'], o['config']['step_launch'], o['final_loss'])
    except Exception as e: print(f, 'ERR', e)
PY
echo R3P_DONE
