#!/bin/bash
# GPU call 23: block-tile shape of the plane trunk, A/B inside the bench step on ONE box (alternating runs)
set -u
OUT=gpurun_out/r03_c23; mkdir -p $OUT
for rep in 1 2; do for sh in 0 1; do
  MH_PLCONV_SHAPE=$sh timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_shape${sh}_$rep.json
  python -c "
import json; d=json.loads(open('$OUT/bench_shape${sh}_$rep.json').read()); t=d['roofline']['trunk_only']; print('shape $sh rep $rep:', round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms; calib', round(d['calibration']['plane_gemm_4096_tflops'],1), '; trunk', round(t['tflops'],1), 'TF', round(t['ms_per_step'],2), 'ms; gemm', round(d['roofline_gemm']['ms_per_step'],2), 'ms')"
done; done
timeout 600 python -m pytest tests/test_gpu_sgdet.py -x -q -s -k "cfg5" > $OUT/cfg5.log 2>&1; grep -E "^cfg5|passed|failed|Error" $OUT/cfg5.log | tail -12 | cut -c1-220
