#!/bin/bash
# GPU call 30 (round 6): the host's run-ahead bound (FusedClipSGD.max_ahead, MOTIFS_MAX_AHEAD) against the cluster of +1.2 ms steps of a young
# process: 20 timed steps after 5 warm-up steps, bounds 1 / 2 / 3 / 4, three rounds
set -u
OUT=gpurun_out/r06_c30; mkdir -p $OUT
show() { python - "$1" "$2" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('%-8s' % sys.argv[2], round(d['value'],1), 'p50', d.get('ms_per_step_p50'), 'host p50/max', d['step_ms']['host_p50'], d['step_ms']['host_max'], d['step_ms']['gpu_per_step'])
PY
}
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --h2d-steps 0"
for rep in 1 2 3; do for a in 4 2 1 3; do MOTIFS_MAX_AHEAD=$a timeout 200 $B > $OUT/a${a}_$rep.json 2>/dev/null; show $OUT/a${a}_$rep.json ahead$a; done; done
