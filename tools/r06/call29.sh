#!/bin/bash
# GPU call 29 (round 6): allocator activity during the timed steps (are the +1 ms steps of a young process device allocations?)
# (MOTIFS_BENCH_ALLOC_TRACE existed in bench.py for this call only: no device allocation happens in the timed steps; removed again)
set -u
OUT=gpurun_out/r06_c29; mkdir -p $OUT
MOTIFS_BENCH_ALLOC_TRACE=1 timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --h2d-steps 0 > $OUT/trace.json 2> $OUT/trace.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_c29/trace.json').read().strip().splitlines()[-1])
print(round(d['value'],1)); 
for t, a in zip(d['step_ms']['gpu_per_step'], d['alloc_trace']): print(t, a)
PY
