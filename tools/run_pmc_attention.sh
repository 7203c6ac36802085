#!/bin/bash
# PMC passes of the encoder layer's attention launches (tools/attn_bench.py): forward and the three backward launches
OUT=gpurun_out/pmc_attn
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
for kind in fwd bwd; do
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace --output-format csv -d $OUT/${kind}_a -o p -- python3 tools/attn_bench.py $kind enc 10 > $OUT/${kind}_a.log 2>&1 || { tail -5 $OUT/${kind}_a.log; exit 1; }
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/${kind}_b -o p -- python3 tools/attn_bench.py $kind enc 10 > $OUT/${kind}_b.log 2>&1 || { tail -5 $OUT/${kind}_b.log; exit 1; }
done
python3 tools/pmc_summary.py $OUT/r04_pmc_attention.json attn_fwd_enc=$OUT/fwd_a,$OUT/fwd_b attn_bwd_enc=$OUT/bwd_a,$OUT/bwd_b > $OUT/pmc.log 2>&1 || tail -5 $OUT/pmc.log
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/pmc_attn/r04_pmc_attention.json'))
for k,v in d['probes'].items(): print(k, v['kernels'], v.get('avg_dispatch_us_under_profiler'), v['derived'])
PY
find $OUT -name "*.csv" -delete; find $OUT -name "*.db" -delete
echo PMC_ATTN_DONE
