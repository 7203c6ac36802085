#!/bin/bash
# PMC passes of the 3x3 kernels on the 40x40x256 layer at batch 16: the halo-resident kernel (hconv.hip) next to the im2col one (BDETR_HCONV=0)
set -o pipefail
OUT=gpurun_out/pmc_r3
mkdir -p $OUT
export TMPDIR=/tmp
G1="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA"
G2="SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
for which in fwd dgrad; do
  for h in 1 0; do
    BDETR_HCONV=$h rocprofv3 --pmc $G1 --kernel-trace --output-format csv -d $OUT/${which}_h${h}_a -o p -- python3 tools/p16_pmc_probe.py $which 40 256 256 3 > $OUT/${which}_h${h}_a.log 2>&1 || { tail -5 $OUT/${which}_h${h}_a.log; exit 1; }
    BDETR_HCONV=$h rocprofv3 --pmc $G2 --kernel-trace --output-format csv -d $OUT/${which}_h${h}_b -o p -- python3 tools/p16_pmc_probe.py $which 40 256 256 3 > $OUT/${which}_h${h}_b.log 2>&1 || { tail -5 $OUT/${which}_h${h}_b.log; exit 1; }
  done
done
python tools/pmc_summary.py $OUT/r03_pmc_3x3_40x40x256.json \
  fwd_halo=$OUT/fwd_h1_a,$OUT/fwd_h1_b fwd_im2col=$OUT/fwd_h0_a,$OUT/fwd_h0_b dgrad_halo=$OUT/dgrad_h1_a,$OUT/dgrad_h1_b dgrad_im2col=$OUT/dgrad_h0_a,$OUT/dgrad_h0_b
find $OUT -name "*kernel_trace.csv" -delete
echo PMC_DONE
