#!/bin/bash
OUT=gpurun_out/r5e
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 1100 python3 -m pytest tests/test_precision_gpu.py tests/test_rowchain_gpu.py tests/test_stem_gpu.py tests/test_training_gpu.py -x -q -m gpu > $OUT/tests.log 2>&1; rc=$?; tail -12 $OUT/tests.log | cut -c1-500; echo "tests rc=$rc"
echo R5E_DONE
