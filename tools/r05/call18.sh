#!/bin/bash
# GPU call 18 (round 5; run twice: the first run found a shadowed variable in train_epoch): models/train_rels.py -m sgdet as a subprocess from a detector checkpoint, the detector stage two batches ahead
set -u
OUT=gpurun_out/r05_c18; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_baselines.py -x -q -m gpu -s -k "relation_driver" > $OUT/tests.log 2>&1; grep -E "passed|failed|rror|detector stage|R@|overall" $OUT/tests.log | tail -12 | cut -c1-300
