#!/bin/bash
# GPU call 10 (round 5): frozen 1x1 convolutions of the ResNet trunk on the plane engine with cached weight images -- the ResNet
# parity tests (blocks, stated-size trunk against the float64 floor, cfg4 at its size) and the cfg4 bench row A/B
set -u
OUT=gpurun_out/r05_c10; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_configs.py -x -q -m gpu -s -k "resnet or cfg4" > $OUT/tests.log 2>&1; grep -E "passed|failed|vs the float64 oracle" $OUT/tests.log | tail -4 | cut -c1-300
row() { python -c "
import json; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', round(d['value'],1), d['unit'], round(d['ms_per_step'],2), 'ms', {k:(round(v['tflops'],1), round(v['ms_per_step'],2), v['launches']) for k,v in d['kernels'].items()}, d['hbm_kernels'].get('make_planes',{}).get('ms_per_step'))" 2>&1 | cut -c1-300; }
timeout 300 python bench.py --config cfg4 --steps 10 --warmup 3 --no-cpu-baseline 2>$OUT/bench_cfg4.err | tail -1 > $OUT/bench_cfg4.json; row $OUT/bench_cfg4.json cfg4_planes
MOTIFS_RESNET_1X1=small timeout 300 python bench.py --config cfg4 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_cfg4_small.json; row $OUT/bench_cfg4_small.json cfg4_small
timeout 300 python bench.py --config cfg4 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_cfg4_b.json; row $OUT/bench_cfg4_b.json cfg4_planes_b
