#!/bin/bash
# 2,000-step soaks at the final state: graph replay and eager enqueue (loss trajectory, step time, allocator high-water marks)
OUT=gpurun_out/soak_r5
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 400 python3 tools/soak.py 2000 split graph > $OUT/r05_soak_2000steps_graph_final.txt 2>&1 || { tail -5 $OUT/r05_soak_2000steps_graph_final.txt; exit 1; }
tail -4 $OUT/r05_soak_2000steps_graph_final.txt
timeout -k 10 400 python3 tools/soak.py 2000 split > $OUT/r05_soak_2000steps_eager_final.txt 2>&1 || { tail -5 $OUT/r05_soak_2000steps_eager_final.txt; exit 1; }
tail -4 $OUT/r05_soak_2000steps_eager_final.txt
echo SOAK_DONE
