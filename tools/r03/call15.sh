#!/bin/bash
set -u
OUT=gpurun_out/r03_c15; mkdir -p $OUT
timeout 200 python tools/r03/diag8_pattern.py 2>&1 | grep -E "TRIAL|   roi|Error|error" > $OUT/diag8.log; cut -c1-600 $OUT/diag8.log
timeout 300 python bench.py --config cfg3 --steps 10 --warmup 3 --host-profile > $OUT/bench_cfg3.log 2> $OUT/bench_cfg3_host.txt; tail -1 $OUT/bench_cfg3.log | cut -c1-300; grep -A24 "Ordered by" $OUT/bench_cfg3_host.txt | cut -c1-160
timeout 600 python -m pytest tests/test_gpu_sgdet.py -x -q 2>&1 | tail -5 | cut -c1-300
