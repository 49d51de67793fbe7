#!/bin/bash
# GPU call 28 (round 6): do the +1 ms steps of a young process go away when the GPU is loaded for a few hundred ms before the warm-up?
# (the --preheat-ms option of bench.py existed for this call only: it changed nothing and was removed again)
set -u
OUT=gpurun_out/r06_c28; mkdir -p $OUT
show() { python - "$1" "$2" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('%-14s' % sys.argv[2], round(d['value'],1), 'p50', d.get('ms_per_step_p50'), 'preheat', d.get('preheat_ms'), d['step_ms']['gpu_per_step'])
PY
}
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --h2d-steps 0"
for rep in 1 2; do
  timeout 200 $B > $OUT/p0_$rep.json 2>/dev/null; show $OUT/p0_$rep.json none
  timeout 200 $B --preheat-ms 400 > $OUT/p400_$rep.json 2>/dev/null; show $OUT/p400_$rep.json preheat400
  timeout 200 $B --preheat-ms 1500 > $OUT/p1500_$rep.json 2>/dev/null; show $OUT/p1500_$rep.json preheat1500
done
