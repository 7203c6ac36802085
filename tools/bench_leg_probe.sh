#!/bin/bash
# Which secondary leg of bench.py perturbs the fp32-grade leg?  Prints headline, value_fp32_grade (and its step count) per flag set.
for flags in "--steps 40 --warmup 5" "--no-cpu-baseline" "--no-roofline" "--no-batch32" ""; do
  python bench.py $flags 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); g=d['value_fp32_grade']; print('[$flags]', d['value'], g['value'], g['steps'], g['with_exact_fp32_backward']['value'], d['value_fp32_policy']['value'])"
done
