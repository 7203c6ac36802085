#!/bin/bash
OUT=gpurun_out/fulltests
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 1100 python -m pytest tests/ -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "rc=$?"; tail -15 $OUT/pytest_gpu.log | cut -c1-400
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $OUT/smoke.log
echo FULLTESTS_DONE
