#!/bin/bash
# GPU call 6 (round 5): the scalar-cache stem with ONE pixel per thread (27 KB of LDS: five blocks per CU instead of two), bitwise test,
# A/B in the step; the host run-ahead bound (MOTIFS_MAX_AHEAD 8 / 4 / 2): do the sporadic 17 ms steps of the first dozen timed steps
# come from the unthrottled host?
set -u
OUT=gpurun_out/r05_c6; mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "stem or trunk" > $OUT/ops_tests.log 2>&1; grep -E "passed|failed|rror" $OUT/ops_tests.log | tail -3 | cut -c1-300
show() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    g=d['step_ms']['gpu_per_step']
    print(sys.argv[1].split('/')[-1], round(d['value'],1), d['unit'], 'ms', round(d['ms_per_step'],2), 'p50', d.get('ms_per_step_p50'), 'unmetered', round(d['unmetered']['value'],1),
          'stem', round(d['hbm_kernels'].get('stem_to_image',{}).get('ms_per_step',0),3), 'cal', round(d['calibration']['plane_gemm_4096_tflops']), 'first12 mean', round(sum(g[:12])/12,2), 'last8 mean', round(sum(g[12:])/8,2))
except Exception as e:
    print(sys.argv[1], 'unreadable', e)
PY
}
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline"
timeout 200 $B > $OUT/bench_scalar.json 2> $OUT/bench_scalar.err; show $OUT/bench_scalar.json
MOTIFS_STEM=lds timeout 200 $B > $OUT/bench_lds.json 2> $OUT/bench_lds.err; show $OUT/bench_lds.json
MOTIFS_MAX_AHEAD=4 timeout 200 $B > $OUT/bench_ahead4.json 2> $OUT/bench_ahead4.err; show $OUT/bench_ahead4.json
MOTIFS_MAX_AHEAD=2 timeout 200 $B > $OUT/bench_ahead2.json 2> $OUT/bench_ahead2.err; show $OUT/bench_ahead2.json
timeout 200 $B > $OUT/bench_scalar_b.json 2> $OUT/bench_scalar_b.err; show $OUT/bench_scalar_b.json
