#!/bin/bash
# GPU call 22 (round 6): fused FrequencyBias add: tests, cfg1 / cfg2 parity, bench A/B; then what a metered step costs the steps around it
# (per-step GPU times at --meter-every 10 / 5 / 1000)
set -u
OUT=gpurun_out/r06_c22; mkdir -p $OUT; R=$PWD
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "frequency_bias or pair_product" > $OUT/tests_ops.log 2>&1; grep -E "passed|failed|rror" $OUT/tests_ops.log | tail -4 | cut -c1-300
timeout 2400 python -m pytest tests/test_gpu_configs.py tests/test_gpu_model.py -x -q -m gpu > $OUT/tests_model.log 2>&1; grep -E "passed|failed|rror" $OUT/tests_model.log | tail -4 | cut -c1-300
show() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], round(d['value'],1), d['unit'], 'ms', round(d['ms_per_step'],2), 'p50', d.get('ms_per_step_p50'), 'cal', round(d['calibration']['plane_gemm_4096_tflops']), 'unmetered', round((d.get('unmetered') or {}).get('value', 0), 1),
          'per step', d['step_ms']['gpu_per_step'])
except Exception as e:
    print(sys.argv[1], 'unreadable', e)
PY
}
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline"
timeout 200 $B > $OUT/bench_new.json 2> $OUT/bench_new.err; show $OUT/bench_new.json
MOTIFS_PAIR_PRODUCT=0 timeout 200 $B > $OUT/bench_off.json 2> /dev/null; show $OUT/bench_off.json
timeout 200 $B > $OUT/bench_new_b.json 2> /dev/null; show $OUT/bench_new_b.json
MOTIFS_PAIR_PRODUCT=0 timeout 200 $B > $OUT/bench_off_b.json 2> /dev/null; show $OUT/bench_off_b.json
timeout 200 $B --meter-every 1000 > $OUT/bench_m1000.json 2> /dev/null; show $OUT/bench_m1000.json
timeout 200 $B --meter-every 5 > $OUT/bench_m5.json 2> /dev/null; show $OUT/bench_m5.json
timeout 200 $B --meter-every 1000 > $OUT/bench_m1000_b.json 2> /dev/null; show $OUT/bench_m1000_b.json
