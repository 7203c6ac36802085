#!/bin/bash
OUT=gpurun_out/r4e
rm -rf $OUT; mkdir -p $OUT
P=tools/probes/graph_replay_repro
for args in "16 40 12 x 409600 0" "16 40 12 x 409600 8" "16 40 12 side 409600 8" "16 10 12 x 1638400 8" "31 60 18 side 409600 8" "16 40 12 x 100000 8"; do
  n=$(echo $args | tr ' ' '_')
  DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 timeout -k 5 120 $P $args > $OUT/on_$n.log 2>&1; echo "rc=$? on $args"; grep -h "differ\|REPRO" $OUT/on_$n.log
done
DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 timeout -k 5 120 $P 16 40 12 x 409600 8 > $OUT/off.log 2>&1; echo "rc=$? off"; grep -h "differ\|REPRO" $OUT/off.log
echo R4E_DONE
