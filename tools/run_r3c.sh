#!/bin/bash
# halo-resident 3x3 backward-data kernel (hconv.hip): correctness, then per-layer timing with and without it; the tests that failed in r3a
set -o pipefail
OUT=gpurun_out/r3c
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_p16_gpu.py -q 2>&1 | tee $OUT/tests_p16.log | tail -15
rc=${PIPESTATUS[0]}
if [ $rc -ge 124 ]; then echo "pytest killed"; exit 1; fi
if [ $rc -ne 0 ]; then echo "P16 TESTS FAILED"; fi
BDETR_HCONV=0 timeout -k 10 300 python tools/p16_bench.py 16 p16 > $OUT/pb_nohconv.log 2>&1 || { tail -5 $OUT/pb_nohconv.log; exit 1; }
timeout -k 10 300 python tools/p16_bench.py 16 p16 > $OUT/pb_hconv.log 2>&1 || { tail -5 $OUT/pb_hconv.log; exit 1; }
paste -d'\n' $OUT/pb_nohconv.log $OUT/pb_hconv.log | grep -E "3x3|per step"
timeout -k 10 600 python -m pytest tests/test_dp_gpu.py tests/test_model_gpu.py -q -s -k "rccl or resnet101" 2>&1 | tee $OUT/tests_fix.log | tail -30
echo R3C_DONE tests_rc=$rc
