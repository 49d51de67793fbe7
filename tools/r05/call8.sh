#!/bin/bash
# GPU call 8 (round 5): bench.py of the final tree -- the driver-style run and the flag-less default run (with the CPU baseline)
set -u
OUT=gpurun_out/r05_c8; mkdir -p $OUT
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_c8/bench.json').read().strip().splitlines()[-1])
print('cfg2', round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'unmetered', round(d['unmetered']['value'],1), d['main_stream_segments'], d['metered_steps'], d['roofline']['dominant_class'], round(d['roofline']['frac'],3))
PY
tail -n 2 $OUT/bench.err | cut -c1-300
( time timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err ) 2>&1 | grep real; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_c8/bench_default.json').read().strip().splitlines()[-1])
print('default', round(d['value'],1), 'img/s', d['steps'], d['warmup'], 'cpu', d['cpu_baseline']['value'], d['cpu_baseline']['sample'][:80])
PY
