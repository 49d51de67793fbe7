#!/bin/bash
# GPU call 27 (round 6): the context stream's priority now that the main stream never waits for it (high = shipped, MOTIFS_SIDE_PRIORITY=0 = default priority)
set -u
OUT=gpurun_out/r06_c27; mkdir -p $OUT
show() { python - "$1" "$2" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('%-14s' % sys.argv[2], round(d['value'],1), 'p50', d.get('ms_per_step_p50'), 'cal', round(d['calibration']['plane_gemm_4096_tflops']), 'unmetered', round((d.get('unmetered') or {}).get('value', 0), 1),
      'seg', {k: round(v, 2) for k, v in d['main_stream_segments'].items() if k.endswith('_ms')})
PY
}
B="python bench.py --steps 24 --warmup 8 --no-cpu-baseline"
for rep in 1 2; do
  timeout 200 $B > $OUT/high_$rep.json 2>/dev/null; show $OUT/high_$rep.json high
  MOTIFS_SIDE_PRIORITY=0 timeout 200 $B > $OUT/normal_$rep.json 2>/dev/null; show $OUT/normal_$rep.json normal
done
