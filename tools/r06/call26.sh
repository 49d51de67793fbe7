#!/bin/bash
# GPU call 26 (round 6): GT rois / image indices assembled on the host, one launch for both BatchNorm counters: parity tests + bench
set -u
OUT=gpurun_out/r06_c26; mkdir -p $OUT
timeout 2400 python -m pytest tests/test_gpu_configs.py tests/test_gpu_model.py tests/test_gpu_sgdet.py tests/test_gpu_baselines.py tests/test_gpu_dist.py -x -q -m gpu > $OUT/tests_model.log 2>&1; grep -E "passed|failed|rror" $OUT/tests_model.log | tail -3 | cut -c1-300
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log | cut -c1-200
show() { python - "$1" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], round(d['value'],1), d['unit'], 'ms', round(d['ms_per_step'],2), 'p50', d.get('ms_per_step_p50'), 'cal', round(d['calibration']['plane_gemm_4096_tflops']), 'unmetered', round((d.get('unmetered') or {}).get('value', 0), 1),
      'seg', {k: round(v, 2) for k, v in d['main_stream_segments'].items() if k.endswith('_ms')})
PY
}
B="python bench.py --steps 24 --warmup 8 --no-cpu-baseline"
timeout 200 $B > $OUT/bench_new.json 2> $OUT/bench_new.err; show $OUT/bench_new.json
timeout 200 $B > $OUT/bench_new_b.json 2> /dev/null; show $OUT/bench_new_b.json
for c in cfg1 cfg5; do timeout 400 python bench.py --config $c --steps 10 --warmup 3 2>/dev/null | tail -1 > $OUT/bench_$c.json; python -c "
import json; d=json.loads(open('$OUT/bench_$c.json').read()); print('$c', round(d['value'],1), d['unit'], round(d['ms_per_step'],2), 'ms')"; done
