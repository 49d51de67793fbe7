#!/bin/bash
# GPU call 16 (round 5): how far ahead the detector stage should run (MOTIFS_DETECT_AHEAD = batches in flight), one or two workers
set -u
OUT=gpurun_out/r05_c16; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_sgdet.py -x -q -m gpu -s -k "ahead" > $OUT/tests.log 2>&1; grep -E "passed|failed|rror|in-line vs" $OUT/tests.log | tail -3 | cut -c1-300
row() { python -c "
import json; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', round(d['value'],1), d['unit'], round(d['ms_per_step'],2), 'ms', d['config'].get('detector_stage'))" 2>&1 | cut -c1-300; }
run() { # name config env...
  n=$1; c=$2; shift 2
  env "$@" timeout 200 python bench.py --config $c --steps 24 --warmup 6 --no-cpu-baseline 2>$OUT/$n.err | tail -1 > $OUT/bench_$n.json; row $OUT/bench_$n.json $n; }
run cfg3_d1 cfg3 MOTIFS_DETECT_AHEAD=1
run cfg3_d2 cfg3 MOTIFS_DETECT_AHEAD=2
run cfg3_d3 cfg3 MOTIFS_DETECT_AHEAD=3
run cfg3_d2_w2 cfg3 MOTIFS_DETECT_AHEAD=2 MOTIFS_AHEAD_WORKERS=2
run cfg3_d3_w2 cfg3 MOTIFS_DETECT_AHEAD=3 MOTIFS_AHEAD_WORKERS=2
run cfg3_d2_p0 cfg3 MOTIFS_DETECT_AHEAD=2 MOTIFS_AHEAD_PRIORITY=0
run cfg3_d0 cfg3 MOTIFS_DETECT_AHEAD=0
run cfg5_d1 cfg5 MOTIFS_DETECT_AHEAD=1
run cfg5_d2 cfg5 MOTIFS_DETECT_AHEAD=2
run cfg5_d2_w2 cfg5 MOTIFS_DETECT_AHEAD=2 MOTIFS_AHEAD_WORKERS=2
run cfg1_d2 cfg1 MOTIFS_DETECT_AHEAD=2
tail -n 3 $OUT/cfg3_d2_w2.err | cut -c1-300
