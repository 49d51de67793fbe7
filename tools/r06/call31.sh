#!/bin/bash
# GPU call 31 (round 6): run-ahead bound 1 against 4 on longer runs and on the other training configurations
set -u
OUT=gpurun_out/r06_c31; mkdir -p $OUT
show() { python - "$1" "$2" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('%-14s' % sys.argv[2], round(d['value'],1), 'ms', round(d['ms_per_step'],2), 'p50', d.get('ms_per_step_p50'))
PY
}
for a in 1 4 1 4; do MOTIFS_MAX_AHEAD=$a timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --h2d-steps 0 > $OUT/cfg2_a${a}.json 2>/dev/null; show $OUT/cfg2_a${a}.json cfg2_60_ahead$a; done
for c in cfg3 cfg4 recipe; do for a in 1 4 1 4; do MOTIFS_MAX_AHEAD=$a timeout 400 python bench.py --config $c --steps 12 --warmup 4 2>/dev/null | tail -1 > $OUT/${c}_a$a.json; python -c "
import json; d=json.loads(open('$OUT/${c}_a$a.json').read()); print('$c ahead$a', round(d['value'],1), d['unit'], round(d['ms_per_step'],2), 'ms')"; done; done
