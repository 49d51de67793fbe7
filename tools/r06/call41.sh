#!/bin/bash
# call 41: the loss node's test with the label guard
mkdir -p gpurun_out/r06_c41
timeout 70 python -m pytest tests/test_gpu_ops.py -q -x -k "cross_entropy_pair or pair_product or frequency_bias" > gpurun_out/r06_c41/t.txt 2>&1
grep -E "passed|failed|Error" gpurun_out/r06_c41/t.txt | tail -3
