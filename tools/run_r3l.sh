#!/bin/bash
OUT=gpurun_out/r3l
mkdir -p $OUT
export BDETR_CXXFLAGS=-DBDETR_SGEMM_DIAG
python -m boosted_detr_amd.build --force > $OUT/build.log 2>&1 || { tail -5 $OUT/build.log; exit 1; }
for dbg in 0 1 2 8 16 3; do
  BDETR_SGEMM_DBG=$dbg timeout -k 10 120 python tools/epi_probe.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/epi.log
done
echo R3L_DONE
