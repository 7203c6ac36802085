#!/bin/bash
OUT=gpurun_out/r4m
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_p16_gpu.py -x -q -k "masked or bn_" > $OUT/pytest_a.log 2>&1; echo "rc=$?"; tail -3 $OUT/pytest_a.log | cut -c1-300
timeout -k 10 900 python -m pytest tests/test_model_gpu.py tests/test_fullsize_gpu.py tests/test_training_gpu.py -x -q > $OUT/pytest_model.log 2>&1; echo "rc=$?"; tail -5 $OUT/pytest_model.log | cut -c1-300
B="--steps 30 --warmup 3 --no-cpu-baseline --no-configs2 --no-batch32 --no-fp32-policy --no-roofline"
for f in 1 0 1 0; do
BDETR_BN_FUSE2=$f timeout -k 10 300 python bench.py $B > $OUT/bench_f$f.json 2> $OUT/bench_f$f.err; python -c "
import json
o=json.loads(open('$OUT/bench_f$f.json').read().strip().split('\n')[-1]); print('BN_FUSE2=$f', o['value'], o['ms_per_step'], o['final_loss'])"
done
B="--no-cpu-baseline --no-roofline --no-batch32 --no-fp32-policy --no-configs2"
BDETR_SIDE_STREAM=0 timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/serial -- python3 bench.py --steps 5 --warmup 2 --no-graph $B > $OUT/serial.log 2>&1; echo "rc=$?"
find $OUT/serial -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_serial.csv \;
find $OUT -name "*kernel_trace.csv" -delete
python tools/kstats.py $OUT/kernel_stats_serial.csv 7 0.5 | grep -i "colreduce\|total\|{" 
echo R4M_DONE
