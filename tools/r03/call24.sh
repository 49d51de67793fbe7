#!/bin/bash
# GPU call 24: the whole -m gpu suite + smoke on the final tree; the two single-image bench rows (trunk engine changed for b = 1)
set -u
OUT=gpurun_out/r03_c24; mkdir -p $OUT
timeout 1200 python -m pytest tests -q -m gpu > $OUT/gpu_suite.log 2>&1; tail -4 $OUT/gpu_suite.log | cut -c1-220
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | cut -c1-200
for c in cfg1 cfg5; do timeout 200 python bench.py --config $c --steps 10 --warmup 3 2>/dev/null | tail -1 > $OUT/bench_$c.json; python -c "
import json; d=json.loads(open('$OUT/bench_$c.json').read()); print('$c', round(d['value'],1), d['unit'], round(d['ms_per_step'],2), 'ms', d.get('kernels'))"; done
