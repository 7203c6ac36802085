#!/bin/bash
# Round-3 profile set at HEAD (BDETR_COMMIT names the commit of the snapshot; the GPU box has no .git).  Summaries land in
# gpurun_out/profiles_r3/ and are copied into profiles/ afterwards.
set -o pipefail
OUT=gpurun_out/profiles_r3
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
B="--no-cpu-baseline --no-roofline --no-batch32 --no-fp32-policy"
python bench.py --steps 40 --warmup 5 > $OUT/r03_bench_line.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
python bench.py --steps 40 --warmup 5 --no-graph --no-cpu-baseline --no-roofline --no-batch32 --no-fp32-policy > $OUT/r03_bench_line_eager.json 2> $OUT/bench_eager.err || { tail -20 $OUT/bench_eager.err; exit 1; }
BDETR_SIDE_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/serial -- python3 bench.py --steps 5 --warmup 2 --no-graph $B > $OUT/serial.log 2>&1 || { tail -20 $OUT/serial.log; exit 1; }
find $OUT/serial -name "*kernel_stats.csv" -exec cp {} $OUT/r03_kernel_stats_serial.csv \;
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/side -- python3 bench.py --steps 5 --warmup 2 --no-graph $B > $OUT/side.log 2>&1 || { tail -20 $OUT/side.log; exit 1; }
find $OUT/side -name "*kernel_stats.csv" -exec cp {} $OUT/r03_kernel_stats_side_stream.csv \;
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/graph -- python3 bench.py --steps 5 --warmup 2 $B > $OUT/graph.log 2>&1 || { tail -20 $OUT/graph.log; exit 1; }
find $OUT/graph -name "*kernel_stats.csv" -exec cp {} $OUT/r03_kernel_stats_graph_replay.csv \;
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 bench.py --steps 2 --warmup 1 --no-graph $B > $OUT/fetch.log 2>&1 || { tail -20 $OUT/fetch.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 bench.py --steps 2 --warmup 1 --no-graph $B > $OUT/write.log 2>&1 || { tail -20 $OUT/write.log; exit 1; }
python tools/hbm_traffic.py $OUT/fetch $OUT/write 3 $OUT/r03_gemm_hbm_traffic.json $OUT/r03_kernel_stats_serial.csv 7 > $OUT/traffic.log 2>&1 || { tail -20 $OUT/traffic.log; exit 1; }
bash tools/run_pmc3x3.sh > $OUT/pmc.log 2>&1 || { tail -20 $OUT/pmc.log; exit 1; }
cp gpurun_out/pmc_r3/r03_pmc_3x3_40x40x256.json $OUT/
find $OUT gpurun_out/pmc_r3 -name "*kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -delete
python tools/kstats.py $OUT/r03_kernel_stats_serial.csv 7 0.25
head -c 600 $OUT/traffic.log; echo
python - <<'PY'
import json
for f in ('r03_bench_line','r03_bench_line_eager'):
    o=json.loads(open(f'gpurun_out/profiles_r3/{f}.json').read().strip().split('\n')[-1])
    print(f, o['value'], o['ms_per_step'], o['config']['step_launch'], o['final_loss'], o.get('roofline',{}) and o['roofline'].get('frac'))
PY
echo PROFILES_DONE
