#!/bin/bash
# Round-5 profile set at HEAD (BDETR_COMMIT names the commit of the snapshot; the GPU box has no .git).  Summaries land in
# gpurun_out/profiles_r5/ and are copied into profiles/ afterwards.
# The runtime switch is exported HERE, before any process starts (a profiler's preloaded tool library initialises the GPU before
# Python runs, so a switch set from inside the process would come too late).  Since round 4 the replay does not depend on it.
set -o pipefail
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
OUT=gpurun_out/profiles_r5
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
B="--no-cpu-baseline --no-roofline --no-batch32 --no-fp32-policy --no-configs2"
python bench.py --steps 40 --warmup 5 > $OUT/r05_bench_line.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
BDETR_PROF_DUMP=$OUT/launches.csv python bench.py --steps 10 --warmup 3 --no-graph --no-cpu-baseline --no-batch32 --no-fp32-policy --no-configs2 > $OUT/bench_launches.json 2> $OUT/bench_launches.err || { tail -20 $OUT/bench_launches.err; exit 1; }
python tools/launch_roofline.py $OUT/launches.csv 10 > $OUT/r05_launch_roofline.txt
python bench.py --steps 40 --warmup 5 --graph $B > $OUT/r05_bench_line_graph.json 2> $OUT/bench_graph.err || { tail -20 $OUT/bench_graph.err; exit 1; }
python bench.py --steps 40 --warmup 5 --no-graph $B > $OUT/r05_bench_line_eager.json 2> $OUT/bench_eager.err || { tail -20 $OUT/bench_eager.err; exit 1; }
BDETR_DETERMINISTIC=1 python bench.py --steps 40 --warmup 5 $B > $OUT/r05_bench_line_deterministic.json 2> $OUT/bench_det.err || { tail -20 $OUT/bench_det.err; exit 1; }
BDETR_SIDE_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/serial -- python3 bench.py --steps 5 --warmup 2 --no-graph $B > $OUT/serial.log 2>&1 || { tail -20 $OUT/serial.log; exit 1; }
find $OUT/serial -name "*kernel_stats.csv" -exec cp {} $OUT/r05_kernel_stats_serial.csv \;
python tools/kernel_by_grid.py "$(find $OUT/serial -name '*kernel_trace.csv' | head -1)" 7 > $OUT/r05_kernel_by_grid_serial.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/side -- python3 bench.py --steps 5 --warmup 2 --no-graph $B > $OUT/side.log 2>&1 || { tail -20 $OUT/side.log; exit 1; }
find $OUT/side -name "*kernel_stats.csv" -exec cp {} $OUT/r05_kernel_stats_side_stream.csv \;
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/graph -- python3 bench.py --steps 5 --warmup 2 --graph $B > $OUT/graph.log 2>&1 || { tail -20 $OUT/graph.log; exit 1; }
find $OUT/graph -name "*kernel_stats.csv" -exec cp {} $OUT/r05_kernel_stats_graph_replay.csv \;
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 bench.py --steps 2 --warmup 1 --no-graph $B > $OUT/fetch.log 2>&1 || { tail -20 $OUT/fetch.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 bench.py --steps 2 --warmup 1 --no-graph $B > $OUT/write.log 2>&1 || { tail -20 $OUT/write.log; exit 1; }
python tools/hbm_traffic.py $OUT/fetch $OUT/write 3 $OUT/r05_gemm_hbm_traffic.json $OUT/r05_kernel_stats_serial.csv 7 > $OUT/traffic.log 2>&1 || { tail -20 $OUT/traffic.log; exit 1; }
# where one step's wall time goes by stream (tools/timeline.py) from the side-stream trace
python tools/timeline.py "$(find $OUT/side -name '*kernel_trace.csv' | head -1)" > $OUT/r05_timeline_side_stream.txt 2>&1 || true
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -delete
python tools/kstats.py $OUT/r05_kernel_stats_serial.csv 7 0.25
python - <<'PY'
import csv, json
rows = list(csv.DictReader(open('gpurun_out/profiles_r5/r05_kernel_stats_serial.csv')))
print('launches per step (serial trace, 7 steps):', round(sum(int(r['Calls']) for r in rows) / 7, 1))
sel = lambda pat: sum(float(r['TotalDurationNs']) for r in rows if any(p in r['Name'] for p in pat)) / 7 / 1e6
print('igemm + attention + LayerNorm + colsum + rowchain ms/step:', round(sel(['igemm_kernel', 'attn_', 'add_drop_ln', 'colsum', 'rowchain_']), 3))
for f in ('r05_bench_line', 'r05_bench_line_graph', 'r05_bench_line_eager', 'r05_bench_line_deterministic'):
    o = json.loads(open(f'gpurun_out/profiles_r5/{f}.json').read().strip().split('\n')[-1])
    print(f, o['value'], o['ms_per_step'], o['config']['step_launch'], o['final_loss'], (o.get('roofline') or {}).get('frac'), (o['config'].get('configs2') or {}).get('value'))
PY
head -c 700 $OUT/traffic.log; echo
echo PROFILES_DONE
