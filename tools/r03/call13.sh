#!/bin/bash
set -u
OUT=gpurun_out/r03_c13; mkdir -p $OUT
for pad in 0 16384; do MH_LDS_PAD=$pad timeout 300 python tools/r03/diag7_pairs.py 2>&1 | grep -E "^PAIR|Error|error" >> $OUT/diag7.log; done; cat $OUT/diag7.log
timeout 300 python bench.py --config cfg3 --steps 10 --warmup 3 --host-profile > $OUT/bench_cfg3.log 2> $OUT/bench_cfg3_host.txt; tail -1 $OUT/bench_cfg3.log | cut -c1-300; grep -A48 "Ordered by" $OUT/bench_cfg3_host.txt | cut -c1-160
