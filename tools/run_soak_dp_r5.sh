#!/bin/bash
# soaks of the DATA-PARALLEL step over a one-rank RCCL communicator: eagerly enqueued (bench.py's default at N > 1) and graph-replayed,
# with the single-process eager step on the same box as the control.  usage: bash tools/run_soak_dp_r5.sh [steps]
N=${1:-1000}
OUT=gpurun_out/soak_dp_r5
rm -rf $OUT; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 300 python3 tools/soak.py 200 split > $OUT/r05_soak_control_200steps_eager.txt 2>&1 || { tail -5 $OUT/r05_soak_control_200steps_eager.txt; exit 1; }
tail -2 $OUT/r05_soak_control_200steps_eager.txt
export BDETR_DP_FORCE=1
timeout -k 10 300 python3 tools/soak.py $N split > $OUT/r05_soak_dp_${N}steps_eager.txt 2>&1 || { tail -5 $OUT/r05_soak_dp_${N}steps_eager.txt; exit 1; }
tail -2 $OUT/r05_soak_dp_${N}steps_eager.txt
timeout -k 10 300 python3 tools/soak.py $N split graph > $OUT/r05_soak_dp_${N}steps_graph.txt 2>&1 || { tail -5 $OUT/r05_soak_dp_${N}steps_graph.txt; exit 1; }
tail -2 $OUT/r05_soak_dp_${N}steps_graph.txt
echo SOAK_DP_DONE
