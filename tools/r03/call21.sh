#!/bin/bash
set -u
OUT=gpurun_out/r03_c21; mkdir -p $OUT
timeout 1500 python -m pytest tests -q -m gpu --durations=8 > $OUT/gpu_suite.log 2>&1; tail -30 $OUT/gpu_suite.log | cut -c1-220
