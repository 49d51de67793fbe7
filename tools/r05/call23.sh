#!/bin/bash
# GPU call 23 (round 5): the trainable ResNet-101 trunk (detector pre-training) -- piece-wise and whole-chain gradients against the oracle
set -u
OUT=gpurun_out/r05_c23; mkdir -p $OUT
timeout 150 python -m pytest tests/test_gpu_model.py -x -q -m gpu -s -k "trainable_resnet" > $OUT/tests.log 2>&1; grep -E "passed|failed|rror|resnet pieces|resnet trunk" $OUT/tests.log | tail -6 | cut -c1-420
