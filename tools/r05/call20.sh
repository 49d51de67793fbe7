#!/bin/bash
# GPU call 20 (round 5): the recorded cfg2 line again after the report change (plane-conv calls split into trunk / small maps, the matrix
# products' fabric traffic from profiles/r05_gemm_traffic_summary.json); library and model code as validated by call 17
set -u
OUT=gpurun_out/r05_c20; mkdir -p $OUT
timeout 400 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_c20/bench.json').read().strip().splitlines()[-1])
r=d['roofline']; c=d['roofline_conv']
print('cfg2', round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'unmetered', round(d['unmetered']['value'],1), 'dominant', r['dominant_class'], round(r['frac'],3), 'traffic', r['traffic'], r.get('traffic_algorithmic'),
      'conv', round(c['frac'],3), 'conv traffic', c['traffic'], 'trunk', c['trunk_only']['launches'], round(c['trunk_only']['frac'],3), 'maps', c['small_maps']['launches'], round(c['small_maps']['frac'],3), 'cpu', d.get('cpu_baseline',{}).get('value'))
PY
tail -2 $OUT/bench.err | cut -c1-200
