#!/bin/bash
# GPU call 24 (round 6): conv arrival counters first in a workspace zeroed once (no memset per launch), one launch to clear both maxima
# arrays of a pair of images: kernel checks, tests, bench A/B
set -u
OUT=gpurun_out/r06_c24; mkdir -p $OUT; R=$PWD
LIB=$R/neural-motifs_amd/csrc/libmotifs_hip.so
( timeout 600 tools/_bin/pl_check $LIB --conv ) > $OUT/conv_check.jsonl 2>&1; grep -c '"ok": true' $OUT/conv_check.jsonl; grep -E '"ok": false|error|summary' $OUT/conv_check.jsonl | cut -c1-300 | head -5
timeout 1200 python -m pytest tests/test_gpu_ops.py -x -q -m gpu > $OUT/tests_ops.log 2>&1; grep -E "passed|failed|rror" $OUT/tests_ops.log | tail -3 | cut -c1-300
timeout 2400 python -m pytest tests/test_gpu_configs.py tests/test_gpu_model.py tests/test_gpu_sgdet.py -x -q -m gpu > $OUT/tests_model.log 2>&1; grep -E "passed|failed|rror" $OUT/tests_model.log | tail -3 | cut -c1-300
show() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], round(d['value'],1), d['unit'], 'ms', round(d['ms_per_step'],2), 'p50', d.get('ms_per_step_p50'), 'cal', round(d['calibration']['plane_gemm_4096_tflops']), 'unmetered', round((d.get('unmetered') or {}).get('value', 0), 1),
          'seg', {k: round(v, 2) for k, v in d['main_stream_segments'].items() if k.endswith('_ms')})
except Exception as e:
    print(sys.argv[1], 'unreadable', e)
PY
}
B="python bench.py --steps 24 --warmup 8 --no-cpu-baseline"
timeout 200 $B > $OUT/bench_new.json 2> $OUT/bench_new.err; show $OUT/bench_new.json
MOTIFS_PLCONV_WS_ZEROED=0 timeout 200 $B > $OUT/bench_memset.json 2> /dev/null; show $OUT/bench_memset.json
timeout 200 $B > $OUT/bench_new_b.json 2> /dev/null; show $OUT/bench_new_b.json
MOTIFS_PLCONV_WS_ZEROED=0 timeout 200 $B > $OUT/bench_memset_b.json 2> /dev/null; show $OUT/bench_memset_b.json
