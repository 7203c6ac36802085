#!/bin/bash
# Round-3 validation pass on the GPU box: full GPU test suite, the default bench line, configs[4] with the mask head + its rocprof summary.
set -o pipefail
OUT=gpurun_out/r3a
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -q -s 2>&1 | tee $OUT/tests.log | tail -60
rc=${PIPESTATUS[0]}
if [ $rc -ge 124 ]; then echo "pytest killed ($rc)"; exit 1; fi
python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err || { tail -30 $OUT/bench.err; exit 1; }
python bench.py --backbone ResNet101 --image 800 --image-w 1333 --queries 300 --batch 8 --panoptic --steps 5 --warmup 2 > $OUT/bench_cfg5_panoptic.json 2> $OUT/bench_cfg5_panoptic.err || { tail -30 $OUT/bench_cfg5_panoptic.err; exit 1; }
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_cfg5 -- python3 bench.py --backbone ResNet101 --image 800 --image-w 1333 --queries 300 --batch 8 --panoptic --steps 3 --warmup 2 --no-roofline > $OUT/prof_cfg5.log 2>&1 || { tail -30 $OUT/prof_cfg5.log; exit 1; }
find $OUT/prof_cfg5 -name "*kernel_stats.csv" -exec cp {} $OUT/cfg5_panoptic_kernel_stats.csv \;
find $OUT/prof_cfg5 -name "*kernel_trace.csv" -delete
echo R3A_DONE tests_rc=$rc
