#!/bin/bash
OUT=gpurun_out/r4l
rm -rf $OUT; mkdir -p $OUT
B="--steps 30 --warmup 3 --no-cpu-baseline --no-configs2 --no-batch32 --no-fp32-policy --no-roofline"
for seg in 10 5 7 14 20; do
BDETR_GRAPH_SEG=$seg timeout -k 10 300 python bench.py $B > $OUT/bench_seg$seg.json 2> $OUT/bench_seg$seg.err; python -c "
import json
o=json.loads(open('$OUT/bench_seg$seg.json').read().strip().split('\n')[-1]); print('SEG=$seg', o['value'], o['ms_per_step'], o['config']['step_launch'])"
done
timeout -k 10 300 python bench.py $B --no-graph > $OUT/bench_eager.json 2> $OUT/bench_eager.err; python -c "
import json
o=json.loads(open('$OUT/bench_eager.json').read().strip().split('\n')[-1]); print('eager', o['value'], o['ms_per_step'], o['config']['step_launch'])"
BDETR_SIDE_STREAM=0 BDETR_GRAPH_SIDE=0 timeout -k 10 300 python bench.py $B > $OUT/bench_noside.json 2> $OUT/bench_noside.err; python -c "
import json
o=json.loads(open('$OUT/bench_noside.json').read().strip().split('\n')[-1]); print('no side stream', o['value'], o['ms_per_step'], o['config']['step_launch'])"
echo R4L_DONE
