#!/bin/bash
# ping-pong 3x3 backward-data kernel: correctness under BDETR_STILE=pp, then per-layer timing against the default tiles; MFMA f16 denormal probe;
# the two tests that failed in r3a
set -o pipefail
OUT=gpurun_out/r3b
mkdir -p $OUT
./tools/probes/mfma_f16_denorm > $OUT/denorm.log 2>&1; cat $OUT/denorm.log
BDETR_STILE=pp timeout -k 10 600 python -m pytest tests/test_p16_gpu.py -q -k "conv_fwd_bwd" 2>&1 | tee $OUT/tests_pp.log | tail -15
rc=${PIPESTATUS[0]}
if [ $rc -ge 124 ]; then echo "pytest killed"; exit 1; fi
timeout -k 10 300 python tools/p16_bench.py 16 p16 > $OUT/pb_default.log 2>&1 || { tail -5 $OUT/pb_default.log; exit 1; }
BDETR_STILE=pp timeout -k 10 300 python tools/p16_bench.py 16 p16 > $OUT/pb_pp.log 2>&1 || { tail -5 $OUT/pb_pp.log; exit 1; }
paste -d'\n' $OUT/pb_default.log $OUT/pb_pp.log | grep -E "3x3|per step"
timeout -k 10 600 python -m pytest tests/test_dp_gpu.py tests/test_model_gpu.py -q -s -k "rccl or resnet101 or forward_outputs or boosted_three or golden" 2>&1 | tee $OUT/tests_fix.log | tail -40
echo R3B_DONE tests_rc=$rc
