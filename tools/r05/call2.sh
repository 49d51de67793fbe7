#!/bin/bash
# GPU call 2 (round 5): the small-product engine with the latency-aware K split and the pipelined fused reduction; the deferred
# optimizer step (own stream, beside the next step's frozen trunk); the segmented gradient reducer on RCCL (world size 1, forced);
# bench A/B: base / deferred optimizer / union-box backward first (MOTIFS_LATE_VR=auto) / both
set -u
OUT=gpurun_out/r05_c2; mkdir -p $OUT; R=$PWD
timeout 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "gemm or small_product or linear" > $OUT/ops_tests.log 2>&1; grep -E "passed|failed|rror" $OUT/ops_tests.log | tail -3 | cut -c1-300
timeout 400 python -m pytest tests/test_gpu_model.py tests/test_gpu_dist.py -x -q -m gpu -k "deferred or two_stream or rccl or reducer" > $OUT/model_tests.log 2>&1; grep -E "passed|failed|rror" $OUT/model_tests.log | tail -3 | cut -c1-300
show() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], round(d['value'],1), d['unit'], 'ms', round(d['ms_per_step'],2), 'p50', d.get('ms_per_step_p50'),
          'gemm', round(d['roofline_gemm']['frac'],3), round(d['roofline_gemm']['ms_per_step'],2), 'conv', round(d['roofline_conv']['frac'],3),
          'cal', round(d['calibration']['plane_gemm_4096_tflops']), 'opt', d['hbm_kernels'].get('fused_clip_sgd',{}).get('ms_per_step'))
except Exception as e:
    print(sys.argv[1], 'unreadable', e)
PY
}
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --h2d-steps 0"
timeout 200 $B --gemm-shapes $OUT/gemm_shapes.jsonl > $OUT/bench_base.json 2> $OUT/bench_base.err; show $OUT/bench_base.json
MOTIFS_OPT_DEFER=1 timeout 200 $B > $OUT/bench_defer.json 2> $OUT/bench_defer.err; show $OUT/bench_defer.json
MOTIFS_LATE_VR=auto timeout 200 $B > $OUT/bench_late.json 2> $OUT/bench_late.err; show $OUT/bench_late.json
MOTIFS_LATE_VR=auto MOTIFS_OPT_DEFER=1 timeout 200 $B > $OUT/bench_late_defer.json 2> $OUT/bench_late_defer.err; show $OUT/bench_late_defer.json
timeout 200 $B > $OUT/bench_base_b.json 2> $OUT/bench_base_b.err; show $OUT/bench_base_b.json
MOTIFS_OPT_DEFER=1 timeout 200 $B > $OUT/bench_defer_b.json 2> $OUT/bench_defer_b.err; show $OUT/bench_defer_b.json
tail -3 $OUT/*.err | cut -c1-300
