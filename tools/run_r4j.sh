#!/bin/bash
OUT=gpurun_out/r4j
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_dp_gpu.py -x -q  > $OUT/pytest_dp.log 2>&1; echo "rc=$?"; tail -30 $OUT/pytest_dp.log
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29553 WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 BDETR_DP_FORCE=1 HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 600 python bench.py --gpus 1 --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-batch32 --no-fp32-policy > $OUT/bench_dp.json 2> $OUT/bench_dp.err; echo "rc=$?"; grep -v "^\[bench\]" $OUT/bench_dp.err | tail -40
echo R4J_DONE
