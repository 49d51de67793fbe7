#!/bin/bash
set -u
OUT=gpurun_out/r03_c20; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_sgdet.py -x -q -s -k "cfg5" > $OUT/cfg5.log 2>&1; grep -E "cfg5|passed|failed|Error" $OUT/cfg5.log | tail -12 | cut -c1-250
