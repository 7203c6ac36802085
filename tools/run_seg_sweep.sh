#!/bin/bash
# hipGraph replay with different segment sizes (side tasks per segment) against the eager step: --graph forces the replay
OUT=gpurun_out/seg_sweep; rm -rf $OUT; mkdir -p $OUT
B="--steps 40 --warmup 5 --no-cpu-baseline --no-roofline --no-batch32 --no-fp32-policy --no-configs2"
python bench.py --no-graph $B > $OUT/eager.json 2> $OUT/eager.err || { tail -5 $OUT/eager.err; exit 1; }
for s in 3 5 10 20; do
  BDETR_GRAPH_SEG=$s python bench.py --graph $B > $OUT/seg_$s.json 2> $OUT/seg_$s.err || { tail -5 $OUT/seg_$s.err; exit 1; }
done
python bench.py --no-graph $B > $OUT/eager2.json 2> $OUT/eager2.err || { tail -5 $OUT/eager2.err; exit 1; }
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/seg_sweep/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], d["ms_per_step"], "ms/step", d["value"], d["config"]["step_launch"][:16])
PY
