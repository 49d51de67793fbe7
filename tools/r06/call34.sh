#!/bin/bash
# GPU call 34 (round 6): Blob.prefetch (next batch's upload on a copy stream): test, bench line with both upload placements
set -u
OUT=gpurun_out/r06_c34; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "prefetched" > $OUT/tests.log 2>&1; grep -E "passed|failed|rror" $OUT/tests.log | tail -3 | cut -c1-300
show() { python - "$1" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
h=d['h2d_inclusive']
print(sys.argv[1].split('/')[-1], round(d['value'],1), 'p50', d.get('ms_per_step_p50'), 'unmetered', round(d['unmetered']['value'],1), 'h2d prefetched', round(h['value'],1), 'h2d inline', round(h['inline']['value'],1))
PY
}
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline"
timeout 300 $B > $OUT/bench_a.json 2> $OUT/bench_a.err; show $OUT/bench_a.json
timeout 300 $B > $OUT/bench_b.json 2> /dev/null; show $OUT/bench_b.json
timeout 600 python - <<'PY' 2>&1 | tail -3
import subprocess, sys, os
# the SGCls driver as a user runs it (synthetic data): the loop now prefetches the next batch
r = subprocess.run([sys.executable, 'neural-motifs_amd/models/train_rels.py', '-m', 'sgcls', '-b', '2', '-nepoch', '1', '-max_iters', '6', '-p', '2', '-ngpu', '1'], capture_output=True, text=True, timeout=500)
print('train_rels rc', r.returncode); print(r.stdout[-400:]); print(r.stderr[-600:] if r.returncode else '')
PY
