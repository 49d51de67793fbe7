#!/bin/bash
# GPU call 5 (round 5): conv1_1 with its weights through the scalar cache (bitwise against the LDS stem, A/B in the step), the LSTM
# weight gradient without its zero fill, bench with quiet_gc in front of the warm-up
set -u
OUT=gpurun_out/r05_c5; mkdir -p $OUT
timeout 400 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "stem or hwlstm or trunk or packed_recurrence or decoder" > $OUT/ops_tests.log 2>&1; grep -E "passed|failed|rror" $OUT/ops_tests.log | tail -3 | cut -c1-300
timeout 400 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "cfg2" > $OUT/cfg2_tests.log 2>&1; grep -E "passed|failed|rror" $OUT/cfg2_tests.log | tail -3 | cut -c1-300
show() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], round(d['value'],1), d['unit'], 'ms', round(d['ms_per_step'],2), 'p50', d.get('ms_per_step_p50'), 'unmetered', round(d['unmetered']['value'],1),
          'stem', d['hbm_kernels'].get('stem_to_image',{}).get('ms_per_step'), 'cal', round(d['calibration']['plane_gemm_4096_tflops']), 'first/last steps', d['step_ms']['gpu_per_step'][:3], d['step_ms']['gpu_per_step'][-3:])
except Exception as e:
    print(sys.argv[1], 'unreadable', e)
PY
}
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline"
timeout 200 $B > $OUT/bench_scalar.json 2> $OUT/bench_scalar.err; show $OUT/bench_scalar.json
MOTIFS_STEM=lds timeout 200 $B > $OUT/bench_lds.json 2> $OUT/bench_lds.err; show $OUT/bench_lds.json
timeout 200 $B > $OUT/bench_scalar_b.json 2> $OUT/bench_scalar_b.err; show $OUT/bench_scalar_b.json
MOTIFS_STEM=lds timeout 200 $B > $OUT/bench_lds_b.json 2> $OUT/bench_lds_b.err; show $OUT/bench_lds_b.json
tail -n 3 $OUT/bench_scalar.err | cut -c1-300
