#!/bin/bash
OUT=gpurun_out/r3m
mkdir -p $OUT
W8=$PWD/tools/probes/libbdetr_w8.so
echo "== 4-wave + EPI split"; timeout -k 10 120 python tools/epi_probe.py 2>&1 | grep -v amdgpu.ids | tee $OUT/epi_w4.log
echo "== 8-wave + EPI split"; BDETR_LIB=$W8 timeout -k 10 120 python tools/epi_probe.py 2>&1 | grep -v amdgpu.ids | tee $OUT/epi_w8.log
timeout -k 10 300 python tools/p16_bench.py 16 p16 > $OUT/p16_w4.log 2>&1; tail -1 $OUT/p16_w4.log
BDETR_LIB=$W8 timeout -k 10 300 python tools/p16_bench.py 16 p16 > $OUT/p16_w8.log 2>&1; tail -1 $OUT/p16_w8.log
timeout -k 10 300 python -m pytest tests/test_p16_gpu.py -x -q 2>&1 | tail -3
BDETR_LIB=$W8 timeout -k 10 300 python -m pytest tests/test_p16_gpu.py -x -q 2>&1 | tail -3
for i in 1 2; do
timeout -k 10 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --no-batch32 --no-fp32-policy > $OUT/bench_w4_$i.json 2> $OUT/bench_w4_$i.err
BDETR_LIB=$W8 timeout -k 10 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --no-batch32 --no-fp32-policy > $OUT/bench_w8_$i.json 2> $OUT/bench_w8_$i.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3m/bench_*.json')):
    try:
        o=json.loads(open(f).read().strip().split('\n')[-1]); print(f, o['value'], o['ms_per_step'], o['config']['step_launch'], o['final_loss'])
    except Exception as e: print(f, 'ERR', e)
PY
echo R3M_DONE
