#!/bin/bash
OUT=gpurun_out/r4k
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 600 python3 tools/hwgrad_ablation.py > $OUT/ablation.log 2>&1; grep "H=" $OUT/ablation.log
echo R4K_DONE
