#!/bin/bash
# GPU call 37 (round 6): the relation driver's two cross-entropy losses as one node (lib/losses.py): test, the drivers as subprocesses,
# bench A/B against MOTIFS_FUSED_LOSS=0
set -u
OUT=gpurun_out/r06_c37; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "cross_entropy_pair" > $OUT/tests_ops.log 2>&1; grep -E "passed|failed|rror" $OUT/tests_ops.log | tail -3 | cut -c1-300
timeout 1500 python -m pytest tests/test_gpu_sgdet.py tests/test_gpu_baselines.py -x -q -m gpu -k "drivers or baseline" > $OUT/tests_drv.log 2>&1; grep -E "passed|failed|rror" $OUT/tests_drv.log | tail -3 | cut -c1-300
show() { python - "$1" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], round(d['value'],1), 'p50', d.get('ms_per_step_p50'), 'cal', round(d['calibration']['plane_gemm_4096_tflops']), 'loss', d['config']['final_loss'],
      'seg', {k: round(v, 2) for k, v in d['main_stream_segments'].items() if k.endswith('_ms')})
PY
}
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline"
timeout 300 $B > $OUT/bench_new.json 2> $OUT/bench_new.err; show $OUT/bench_new.json
MOTIFS_FUSED_LOSS=0 timeout 300 $B > $OUT/bench_off.json 2> /dev/null; show $OUT/bench_off.json
timeout 300 $B > $OUT/bench_new_b.json 2> /dev/null; show $OUT/bench_new_b.json
MOTIFS_FUSED_LOSS=0 timeout 300 $B > $OUT/bench_off_b.json 2> /dev/null; show $OUT/bench_off_b.json
