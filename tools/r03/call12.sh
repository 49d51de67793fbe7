#!/bin/bash
# GPU call 12: kernel-pair race search; cfg3 host profile
set -u
OUT=gpurun_out/r03_c12; mkdir -p $OUT
timeout 300 python tools/r03/diag6_pairs.py 2>&1 | grep -E "^PAIR|Error|error" > $OUT/diag6.log; cat $OUT/diag6.log
timeout 300 python bench.py --config cfg3 --steps 10 --warmup 3 --host-profile > $OUT/bench_cfg3.log 2> $OUT/bench_cfg3_host.txt; tail -1 $OUT/bench_cfg3.log | cut -c1-300; grep -A48 "Ordered by" $OUT/bench_cfg3_host.txt | cut -c1-160
