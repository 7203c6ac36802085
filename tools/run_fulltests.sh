#!/bin/bash
OUT=gpurun_out/fulltests
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 1100 python -m pytest tests/ -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "rc=$?"; tail -15 $OUT/pytest_gpu.log
echo FULLTESTS_DONE
