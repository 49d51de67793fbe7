#!/bin/bash
# GPU call 17 (round 4): the prefetching NMS sweep -- timing against round 3's chained sweep, then the whole -m gpu suite on the final
# tree (NMS sizes up to 12000 boxes and crowded scenes added), smoke, and the cfg3 / cfg2 bench lines
set -u
OUT=gpurun_out/r04_c17; mkdir -p $OUT
timeout 120 python tools/r04/nms_time.py > $OUT/nms_time.jsonl 2> $OUT/nms_time.err; cat $OUT/nms_time.jsonl | cut -c1-400
timeout 1300 python -m pytest tests/ -x -q -m gpu > $OUT/gpu_tests.log 2>&1; tail -2 $OUT/gpu_tests.log | cut -c1-300
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | cut -c1-200
for c in cfg3 cfg1; do timeout 200 python bench.py --config $c --steps 10 --warmup 3 2>/dev/null | tail -1 > $OUT/bench_$c.json; python -c "
import json; d=json.loads(open('$OUT/bench_$c.json').read()); print('$c', round(d['value'],1), d['unit'], round(d['ms_per_step'],2), 'ms')"; done
timeout 300 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; tail -1 $OUT/bench.json | cut -c1-250
