#!/bin/bash
# GPU call 19 (round 4): secondary bench rows on the final tree (mask kernel with scalar-cache column boxes included); cfg3 also without
# the kernel meters (event pairs around ~1000 launches of a metered step)
set -u
OUT=gpurun_out/r04_c19; mkdir -p $OUT
for c in cfg3 cfg5 cfg4 recipe; do timeout 100 python bench.py --config $c --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_$c.json; python -c "
import json; d=json.loads(open('$OUT/bench_$c.json').read()); print('$c', round(d['value'],1), d['unit'], round(d['ms_per_step'],2), 'ms', d.get('ms_per_step_p50'))"; done
timeout 100 python bench.py --config cfg3 --steps 10 --warmup 3 --no-cpu-baseline --meter-every 1000 --h2d-steps 0 2>/dev/null | tail -1 > $OUT/bench_cfg3_unmetered.json; python -c "
import json; d=json.loads(open('$OUT/bench_cfg3_unmetered.json').read()); print('cfg3 unmetered', round(d['value'],1), d['unit'], round(d['ms_per_step'],2), 'ms', d.get('ms_per_step_p50'))"
