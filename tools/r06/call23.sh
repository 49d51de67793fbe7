#!/bin/bash
# GPU call 23 (round 6): the +1 ms steps among the first dozen timed steps: warm-up length, run-ahead bound
set -u
OUT=gpurun_out/r06_c23; mkdir -p $OUT
show() { python - "$1" "$2" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('%-22s' % sys.argv[2], round(d['value'],1), 'p50', d.get('ms_per_step_p50'), 'host p50/max', d['step_ms']['host_p50'], d['step_ms']['host_max'], d['step_ms']['gpu_per_step'])
PY
}
B="python bench.py --steps 24 --no-cpu-baseline --h2d-steps 0"
timeout 200 $B --warmup 5 > $OUT/w5.json 2>/dev/null; show $OUT/w5.json warmup5
timeout 200 $B --warmup 24 > $OUT/w24.json 2>/dev/null; show $OUT/w24.json warmup24
MOTIFS_MAX_AHEAD=2 timeout 200 $B --warmup 5 > $OUT/a2.json 2>/dev/null; show $OUT/a2.json ahead2
MOTIFS_MAX_AHEAD=8 timeout 200 $B --warmup 5 > $OUT/a8.json 2>/dev/null; show $OUT/a8.json ahead8
MOTIFS_MAX_AHEAD=-1 timeout 200 $B --warmup 5 > $OUT/ainf.json 2>/dev/null; show $OUT/ainf.json unbounded
timeout 200 $B --warmup 5 --meter-every 1000 > $OUT/w5m.json 2>/dev/null; show $OUT/w5m.json warmup5_nometer
