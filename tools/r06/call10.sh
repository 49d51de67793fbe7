#!/bin/bash
# GPU call 10 (round 6): the mask tower's 7x7 convolution on the matrix cores (t1::tower_conv1_mfma_*): kernel tests, cfg2 parity,
# bench A/B against MH_TOWER_CONV1=valu, per-kernel durations
set -u
OUT=gpurun_out/r06_c10; mkdir -p $OUT; R=$PWD
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "tower" > $OUT/tests_tower.log 2>&1; grep -E "passed|failed|rror|Mismatch|Max abs|Max rel" $OUT/tests_tower.log | tail -8 | cut -c1-300
MH_TOWER_CONV1=valu timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "tower" > $OUT/tests_tower_valu.log 2>&1; grep -E "passed|failed|rror" $OUT/tests_tower_valu.log | tail -3 | cut -c1-300
timeout 1200 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "cfg2 or cfg1" > $OUT/tests_cfg.log 2>&1; grep -E "passed|failed|rror" $OUT/tests_cfg.log | tail -3 | cut -c1-300
show() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], round(d['value'],1), d['unit'], 'ms', round(d['ms_per_step'],2), 'p50', d.get('ms_per_step_p50'),
          'gemm', round(d['roofline_gemm']['frac'],3), round(d['roofline_gemm']['ms_per_step'],2), 'imgs', round(d['roofline_gemm']['products_on_images']['frac'],3),
          'conv', round(d['roofline_conv']['frac'],3), 'trunk', round(d['roofline']['frac_trunk_only'],3), round(d['roofline_conv']['trunk_only']['ms_per_step'],2), 'cal', round(d['calibration']['plane_gemm_4096_tflops']),
          'seg', {k: round(v, 2) for k, v in d['main_stream_segments'].items() if k.endswith('_ms')})
except Exception as e:
    print(sys.argv[1], 'unreadable', e)
PY
}
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline"
timeout 200 $B > $OUT/bench_new.json 2> $OUT/bench_new.err; show $OUT/bench_new.json
MH_TOWER_CONV1=valu timeout 200 $B > $OUT/bench_valu.json 2> /dev/null; show $OUT/bench_valu.json
timeout 200 $B > $OUT/bench_new_b.json 2> /dev/null; show $OUT/bench_new_b.json
MH_TOWER_CONV1=valu timeout 200 $B > $OUT/bench_valu_b.json 2> /dev/null; show $OUT/bench_valu_b.json
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_cfg2 -- python $R/bench.py --steps 8 --warmup 4 --no-cpu-baseline --h2d-steps 0 --meter-every 1000 > $R/$OUT/prof_cfg2.log 2>&1 )
cp $(ls /tmp/prof_cfg2/*/*kernel_stats.csv | head -1) $OUT/kernel_stats_cfg2.csv 2>/dev/null
grep -E "tower|bn_" $OUT/kernel_stats_cfg2.csv | cut -d, -f1-4 | cut -c1-200
