#!/bin/bash
# GPU call 21 (round 4): the direct first convolution of the mask tower -- kernels against conv2d, tower A/B, step A/B
set -u
OUT=gpurun_out/r04_c21; mkdir -p $OUT
timeout 70 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "tower" > $OUT/tower_tests.log 2>&1; grep -E "passed|failed|rror" $OUT/tower_tests.log | tail -3 | cut -c1-300
for v in direct gemm; do MOTIFS_TOWER_CONV1=$v timeout 45 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --h2d-steps 0 2>/dev/null | tail -1 > $OUT/bench_$v.json; python -c "
import json; d=json.loads(open('$OUT/bench_$v.json').read()); print('$v', round(d['value'],1), d['unit'], round(d['ms_per_step'],2), 'ms p50', d.get('ms_per_step_p50'))"; done
