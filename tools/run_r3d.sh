#!/bin/bash
# unscaled f16 pairs + one accumulator + halo forward: kernel-level tests, precision tests, per-layer timing, then the whole GPU suite and the bench
set -o pipefail
OUT=gpurun_out/r3d
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_p16_gpu.py tests/test_precision_gpu.py tests/test_kernels_gpu.py -q 2>&1 | tee $OUT/tests_kern.log | tail -25
rc=${PIPESTATUS[0]}
if [ $rc -ge 124 ]; then echo "pytest killed"; exit 1; fi
timeout -k 10 300 python tools/p16_bench.py 16 p16 > $OUT/pb.log 2>&1 || { tail -5 $OUT/pb.log; exit 1; }
grep -E "3x3|per step" $OUT/pb.log
if [ $rc -ne 0 ]; then echo "KERNEL TESTS FAILED - stopping"; exit 0; fi
timeout -k 10 1000 python -m pytest tests -m gpu -q 2>&1 | tee $OUT/tests_all.log | tail -30
python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err || { tail -30 $OUT/bench.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3d/bench.json').read().strip().splitlines()[-1])
r=d['roofline']
print('images/s',d['value'],'ms',d['ms_per_step'],'frac',r['frac'],'kernel ms',r['kernel_ms_per_step'])
for k,v in r['by_class'].items(): print(' ',k[:60],v['kernel_ms_per_step'],v['frac_of_mfma_roof'],v['frac_of_hbm_roof'])
print(d['config']['configs3'], d['value_fp32_policy'])
PY
echo R3D_DONE
