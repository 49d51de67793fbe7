#!/bin/bash
# GPU call 24 (round 5): models/train_detector.py -resnet as a subprocess (2 training batches + the validation epoch)
set -u
OUT=gpurun_out/r05_c24; mkdir -p $OUT
timeout 130 python -m pytest tests/test_gpu_baselines.py -x -q -m gpu -k "detector_driver and resnet" > $OUT/tests.log 2>&1; grep -E "passed|failed|Error|error" $OUT/tests.log | tail -6 | cut -c1-400
