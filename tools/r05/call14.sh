#!/bin/bash
# GPU call 14 (round 5): the detector stage one batch ahead (RelModel.detect_ahead: worker thread + own HIP stream) -- equality with
# the in-line order, then the SGDet rows A/B (cfg3 training step, cfg5 evaluation; MOTIFS_DETECT_AHEAD=0 = in line)
set -u
OUT=gpurun_out/r05_c14; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_sgdet.py -x -q -m gpu -k "ahead or cfg3 or train_step_parity" > $OUT/tests.log 2>&1; grep -E "passed|failed|rror" $OUT/tests.log | tail -3 | cut -c1-300
row() { python -c "
import json; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', round(d['value'],1), d['unit'], round(d['ms_per_step'],2), 'ms', d['config'].get('detector_stage'), d['config'].get('dets'), d['config'].get('rows'))" 2>&1 | cut -c1-300; }
for t in a b; do
  timeout 200 python bench.py --config cfg3 --steps 20 --warmup 5 --no-cpu-baseline 2>$OUT/cfg3_ahead_$t.err | tail -1 > $OUT/bench_cfg3_ahead_$t.json; row $OUT/bench_cfg3_ahead_$t.json cfg3_ahead
  MOTIFS_DETECT_AHEAD=0 timeout 200 python bench.py --config cfg3 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_cfg3_inline_$t.json; row $OUT/bench_cfg3_inline_$t.json cfg3_inline
done
timeout 200 python bench.py --config cfg5 --steps 20 --warmup 5 --no-cpu-baseline 2>$OUT/cfg5_ahead.err | tail -1 > $OUT/bench_cfg5_ahead.json; row $OUT/bench_cfg5_ahead.json cfg5_ahead
MOTIFS_DETECT_AHEAD=0 timeout 200 python bench.py --config cfg5 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_cfg5_inline.json; row $OUT/bench_cfg5_inline.json cfg5_inline
tail -n 4 $OUT/cfg3_ahead_a.err | cut -c1-300
