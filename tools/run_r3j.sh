#!/bin/bash
# failing configuration of tools/graph_debug.py without any exported switch (the package sets it at import), the training tests,
# then eager / graph bench A/B on one box
OUT=gpurun_out/r3j
mkdir -p $OUT
run() { echo "== $*" | tee -a $OUT/debug.log; env "$@" timeout -k 10 200 python tools/graph_debug.py 2>&1 | grep -vE "^eager|Warning|warn|amdgpu.ids" | grep -oE "tripped.*|.'after'.*|redoing.*|Error.*" | cut -c1-200 | head -8 | tee -a $OUT/debug.log; }
rm -f $OUT/debug.log
run GUARD=1 GRAPH_ONLY=0
run GUARD=1 GRAPH_ONLY=0
timeout -k 10 900 python -m pytest tests/test_training_gpu.py -x -q -m gpu 2>&1 | tail -15 | tee $OUT/pytest_training.log && \
timeout -k 10 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline > $OUT/bench_eager.json 2> $OUT/bench_eager.err && \
timeout -k 10 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --graph > $OUT/bench_graph.json 2> $OUT/bench_graph.err && \
timeout -k 10 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline > $OUT/bench_eager2.json 2> $OUT/bench_eager2.err && \
timeout -k 10 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --graph > $OUT/bench_graph2.json 2> $OUT/bench_graph2.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3j/bench_*.json')):
    try:
        o=json.loads(open(f).read().strip().split('\n')[-1]); print(f, o['value'], o['ms_per_step'], o['config']['step_launch'], o['range_guard'], o['final_loss'])
    except Exception as e: print(f, 'ERR', e)
PY
echo R3J_DONE
