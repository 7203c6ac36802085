#!/bin/bash
OUT=gpurun_out/r4o
rm -rf $OUT; mkdir -p $OUT
DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 timeout -k 10 600 python3 tools/soak.py 2000 split graph > $OUT/soak_packets_on.log 2>&1; echo "rc=$?"; (head -3; tail -4) < $OUT/soak_packets_on.log | cut -c1-220
echo R4O_DONE
