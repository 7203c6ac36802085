#!/bin/bash
OUT=gpurun_out/r4n
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_fullsize_gpu.py tests/test_model_gpu.py -x -q -s -k "config2_batch2 or split_and_fp32 or replay or gradients" > $OUT/pytest.log 2>&1; echo "rc=$?"; grep -v "^$" $OUT/pytest.log | tail -15 | cut -c1-400
echo R4N_DONE
