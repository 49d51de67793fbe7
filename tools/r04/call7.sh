#!/bin/bash
# GPU call 7 (round 4): full -m gpu suite on the new default dispatch, smoke, bench
set -u
OUT=gpurun_out/r04_c7; mkdir -p $OUT
timeout 900 python -m pytest tests/ -x -q -m gpu > $OUT/gpu_tests.log 2>&1; tail -5 $OUT/gpu_tests.log | cut -c1-300
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | cut -c1-200
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; tail -2 $OUT/bench.err | cut -c1-300
python -c "
import json; d=json.loads(open('$OUT/bench.json').read().strip().split('\n')[-1]); t=d['roofline']['trunk_only']; print('bench', round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms; p50/p90/max', d['ms_per_step_p50'], d['ms_per_step_p90'], d['ms_per_step_max'], '; h2d', d['h2d_inclusive'] and round(d['h2d_inclusive']['ms_per_step'],2), '; calib', round(d['calibration']['plane_gemm_4096_tflops'],1), '; conv', round(d['roofline']['achieved'],1), 'trunk', round(t['tflops'],1), round(t['ms_per_step'],2), 'gemm', round(d['roofline_gemm']['achieved'],1), round(d['roofline_gemm']['ms_per_step'],2)); print(d['step_ms']['gpu_per_step'])"
