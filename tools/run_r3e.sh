#!/bin/bash
set -o pipefail
OUT=gpurun_out/r3e
mkdir -p $OUT
timeout -k 10 1000 python -m pytest tests -m gpu -q 2>&1 | tee $OUT/tests_all.log | tail -30
python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err || { tail -30 $OUT/bench.err; exit 1; }
BDETR_HCONV=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-batch32 --no-fp32-policy > $OUT/bench_nohconv.json 2> $OUT/bench_nohconv.err || { tail -30 $OUT/bench_nohconv.err; exit 1; }
python - <<'PY'
import json
for f in ('bench','bench_nohconv'):
    d=json.loads(open(f'gpurun_out/r3e/{f}.json').read().strip().splitlines()[-1])
    r=d['roofline']
    print(f,'images/s',d['value'],'ms',d['ms_per_step'],'frac',r['frac'],'kernel ms',r['kernel_ms_per_step'])
    for k,v in r['by_class'].items(): print('   ',k[:60],v['kernel_ms_per_step'],v['frac_of_mfma_roof'],v['frac_of_hbm_roof'])
    print('   ',d['config']['configs3'], d['value_fp32_policy'])
PY
echo R3E_DONE
