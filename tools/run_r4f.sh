#!/bin/bash
OUT=gpurun_out/r4f
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_rowchain_gpu.py -q > $OUT/pytest_rowchain.log 2>&1; echo "rc=$?"; tail -30 $OUT/pytest_rowchain.log
echo R4F_DONE
timeout -k 10 900 python -m pytest tests/test_model_gpu.py tests/test_training_gpu.py tests/test_fullsize_gpu.py tests/test_precision_gpu.py -x -q > $OUT/pytest_model.log 2>&1; echo "rc=$?"; tail -15 $OUT/pytest_model.log
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-configs2 > $OUT/bench.json 2> $OUT/bench.err; echo "rc=$?"; tail -3 $OUT/bench.err; python - <<'PY'
import json
o=json.loads(open('gpurun_out/r4f/bench.json').read().strip().split('\n')[-1])
print(o['value'], o['ms_per_step'], o['config']['step_launch'], o['final_loss'], o['roofline']['frac'], o['roofline']['launches_per_step'], o['value_fp32_policy'], o['config']['configs3'])
for k,v in o['roofline']['by_class'].items(): print(k, v)
PY
BDETR_ROWCHAIN=0 timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-configs2 --no-roofline --no-batch32 --no-fp32-policy > $OUT/bench_norc.json 2> $OUT/bench_norc.err; echo "rc=$?"; python -c "
import json
o=json.loads(open('gpurun_out/r4f/bench_norc.json').read().strip().split('\n')[-1]); print('ROWCHAIN=0', o['value'], o['ms_per_step'], o['final_loss'])"
echo R4F2_DONE
