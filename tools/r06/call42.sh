#!/bin/bash
# call 42: SGDet evaluation / training tests on the final tree (decoder refactor in the eval path)
mkdir -p gpurun_out/r06_c42
timeout 115 python -m pytest tests/test_gpu_sgdet.py -q -x > gpurun_out/r06_c42/t.txt 2>&1
grep -E "passed|failed|Error" gpurun_out/r06_c42/t.txt | tail -3
