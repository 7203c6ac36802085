#!/bin/bash
OUT=gpurun_out/r4p
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 600 python3 tools/hconv_tile_ab.py > $OUT/hconv_tile_ab.json 2> $OUT/err.log; echo "rc=$?"; cat $OUT/hconv_tile_ab.json | head -80
echo R4P_DONE
