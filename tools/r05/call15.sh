#!/bin/bash
# GPU call 15 (round 5): detect-ahead equality test (parameters held to the atomics noise of torch's index_add_), the idle time of
# the SGDet step with the detector stage one batch ahead (rocprofv3 kernel trace -> tools/trace_gaps.py), PredCls eval A/B (=all)
set -u
OUT=gpurun_out/r05_c15; mkdir -p $OUT; R=$PWD
timeout 600 python -m pytest tests/test_gpu_sgdet.py -x -q -m gpu -s -k "ahead" > $OUT/tests.log 2>&1; grep -E "passed|failed|rror|in-line vs" $OUT/tests.log | tail -4 | cut -c1-400
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_cfg3 -- python $R/bench.py --config cfg3 --steps 8 --warmup 4 --no-cpu-baseline --meter-every 100 > $R/$OUT/prof_cfg3.log 2>&1 )
T=$(ls /tmp/prof_cfg3/*/*kernel_trace.csv | head -1)
cp $(ls /tmp/prof_cfg3/*/*kernel_stats.csv | head -1) $OUT/kernel_stats_cfg3.csv 2>/dev/null
python tools/trace_gaps.py $T --steps 3 --top 8 > $OUT/trace_gaps_cfg3.txt 2>&1; head -8 $OUT/trace_gaps_cfg3.txt | cut -c1-200
row() { python -c "
import json; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', round(d['value'],1), d['unit'], round(d['ms_per_step'],2), 'ms', d['config'].get('detector_stage'))" 2>&1 | cut -c1-300; }
MOTIFS_DETECT_AHEAD=all timeout 200 python bench.py --config cfg1 --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_cfg1_ahead.json; row $OUT/bench_cfg1_ahead.json cfg1_ahead
timeout 200 python bench.py --config cfg1 --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_cfg1_inline.json; row $OUT/bench_cfg1_inline.json cfg1_inline
