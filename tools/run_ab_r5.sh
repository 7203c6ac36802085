#!/bin/bash
# A/B of one environment switch on the headline bench (alternating runs on one box): tools/run_ab_r5.sh NAME "ENV=VAL" [reps]
OUT=gpurun_out/ab_$1; mkdir -p $OUT
B="--no-cpu-baseline --no-roofline --no-batch32 --no-fp32-policy --no-configs2"
REPS=${3:-2}
for i in $(seq 1 $REPS); do
  python bench.py --steps 40 --warmup 5 $B > $OUT/base_$i.json 2> $OUT/base_$i.err || { tail -5 $OUT/base_$i.err; exit 1; }
  env $2 python bench.py --steps 40 --warmup 5 $B > $OUT/alt_$i.json 2> $OUT/alt_$i.err || { tail -5 $OUT/alt_$i.err; exit 1; }
done
python - "$OUT" "$2" <<'PY'
import json, sys, glob
out, sw = sys.argv[1], sys.argv[2]
val = lambda f: json.loads(open(f).read().strip().splitlines()[-1])["value"]
b = [val(f) for f in sorted(glob.glob(out + "/base_*.json"))]; a = [val(f) for f in sorted(glob.glob(out + "/alt_*.json"))]
print("default:", b, " with", sw + ":", a)
PY
