#!/bin/bash
# does the timed region's length change the per-step figure?  single process and the one-rank RCCL data-parallel path, 10 and 40 timed steps
OUT=gpurun_out/steps_sweep; rm -rf $OUT; mkdir -p $OUT
B="--warmup 5 --no-cpu-baseline --no-roofline --no-batch32 --no-fp32-policy --no-configs2"
for k in 10 40; do
  python bench.py --steps $k $B > $OUT/single_$k.json 2> $OUT/single_$k.err || { tail -5 $OUT/single_$k.err; exit 1; }
  BDETR_DP_FORCE=1 python bench.py --steps $k $B > $OUT/dp_$k.json 2> $OUT/dp_$k.err || { tail -5 $OUT/dp_$k.err; exit 1; }
done
python - <<'PY'
import json
for n in ("single_10", "single_40", "dp_10", "dp_40"):
    d = json.loads(open(f"gpurun_out/steps_sweep/{n}.json").read().strip().splitlines()[-1])
    print(n, d["ms_per_step"], "ms/step", d["value"], "images/s", d["config"]["step_launch"][:12])
PY
