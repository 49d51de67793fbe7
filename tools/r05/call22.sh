#!/bin/bash
# GPU call 22 (round 5): last sanity on the final tree -- smoke + the run-ahead equality test
set -u
OUT=gpurun_out/r05_c22; mkdir -p $OUT
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | cut -c1-200
timeout 300 python -m pytest tests/test_gpu_sgdet.py -x -q -m gpu -k "ahead" > $OUT/tests.log 2>&1; grep -E "passed|failed|rror" $OUT/tests.log | tail -2 | cut -c1-300
