#!/bin/bash
OUT=gpurun_out/r4h
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 python3 tools/rowchain_bench.py > $OUT/rc_bench_pf4.log 2>&1; echo "rc=$? PF=4"; grep "M=" $OUT/rc_bench_pf4.log
for pf in 6 8; do
BDETR_LIB=$PWD/boosted_detr_amd/csrc/_alt/libbdetr_pf$pf.so timeout -k 10 300 python3 tools/rowchain_bench.py > $OUT/rc_bench_pf$pf.log 2>&1; echo "rc=$? PF=$pf"; grep "M=" $OUT/rc_bench_pf$pf.log
done
timeout -k 10 600 python -m pytest tests/test_rowchain_gpu.py -q > $OUT/pytest_rowchain.log 2>&1; echo "rc=$?"; tail -3 $OUT/pytest_rowchain.log
echo R4H_DONE
