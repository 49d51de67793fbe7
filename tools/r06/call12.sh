#!/bin/bash
# GPU call 12 (round 6): tower tests (three forms, kink-aware), cfg2 / cfg4 parity with the un-forced gradient report, then the whole
# -m gpu suite on this tree
set -u
OUT=gpurun_out/r06_c12; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "tower" -s > $OUT/tests_tower.log 2>&1; grep -E "passed|failed|rror|seed" $OUT/tests_tower.log | tail -8 | cut -c1-300
timeout 2400 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "cfg2 or cfg4_resnet_sgcls" -s > $OUT/tests_cfg.log 2>&1; grep -E "passed|failed|rror|worst of" $OUT/tests_cfg.log | tail -8 | cut -c1-300
timeout 2400 python -m pytest tests -x -q -m gpu > $OUT/tests_all.log 2>&1; grep -E "passed|failed|rror" $OUT/tests_all.log | tail -5 | cut -c1-300
