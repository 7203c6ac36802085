#!/bin/bash
OUT=gpurun_out/r5a
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 120 python3 tools/attn_bench.py all all 50 > $OUT/attn_bench.txt 2>&1 || { tail -5 $OUT/attn_bench.txt; exit 1; }
cat $OUT/attn_bench.txt
for kind in fwd; do
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace --output-format csv -d $OUT/${kind}_a -o p -- python3 tools/attn_bench.py $kind enc 10 > $OUT/${kind}_a.log 2>&1 || { tail -5 $OUT/${kind}_a.log; exit 1; }
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/${kind}_b -o p -- python3 tools/attn_bench.py $kind enc 10 > $OUT/${kind}_b.log 2>&1 || { tail -5 $OUT/${kind}_b.log; exit 1; }
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM --kernel-trace --output-format csv -d $OUT/${kind}_c -o p -- python3 tools/attn_bench.py $kind enc 10 > $OUT/${kind}_c.log 2>&1 || { tail -5 $OUT/${kind}_c.log; exit 1; }
done
python3 tools/pmc_summary.py $OUT/pmc_attn.json attn_fwd_enc=$OUT/fwd_a,$OUT/fwd_b,$OUT/fwd_c > $OUT/pmc.log 2>&1 || tail -5 $OUT/pmc.log
cat $OUT/pmc_attn.json
find $OUT -name "*.csv" -delete; find $OUT -name "*.db" -delete
echo R5A_DONE
