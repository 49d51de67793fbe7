#!/bin/bash
set -u
OUT=gpurun_out/r03_c14; mkdir -p $OUT
timeout 300 python tools/r03/diag7_pairs.py 2>&1 | grep -E "^PAIR|Error|error" > $OUT/diag7.log; cat $OUT/diag7.log | cut -c1-200
timeout 240 python tools/r03/diag3_streams.py roi_unconditional_loads 2>&1 | grep VARIANT | tee $OUT/diag3.log
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "roi or RoI" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_sgdet.py -x -q 2>&1 | tail -8 | cut -c1-300
