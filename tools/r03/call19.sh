#!/bin/bash
set -u
OUT=gpurun_out/r03_c19; mkdir -p $OUT; R=$PWD
timeout 300 python -m pytest tests/test_gpu_ops.py -x -q -k "next_to_the_in_loop" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_model.py -x -q -k "two_stream" 2>&1 | tail -6 | cut -c1-300
timeout 600 python -m pytest tests/test_gpu_sgdet.py -x -q -s -k "cfg5" > $OUT/cfg5.log 2>&1; grep -E "cfg5|passed|failed|Error|assert" $OUT/cfg5.log | tail -25 | cut -c1-250
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 > $OUT/bench.log
python -c "
import json,sys
d=json.loads(open('$OUT/bench.log').read()); print('bench', round(d['value'],1), round(d['ms_per_step'],2), 'calib', round(d['calibration']['plane_gemm_4096_tflops'],1), 'conv', round(d['roofline']['achieved'],1), 'trunk', round(d['roofline']['trunk_only']['tflops'],1), round(d['roofline']['trunk_only']['ms_per_step'],2), 'gemm', round(d['roofline_gemm']['achieved'],1), round(d['roofline_gemm']['ms_per_step'],2))"
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof3 -- python $R/bench.py --config cfg3 --steps 6 --warmup 3 > $R/$OUT/prof_cfg3.log 2>&1 )
cp $(ls /tmp/prof3/*/*kernel_stats.csv | head -1) $OUT/cfg3_kernel_stats.csv 2>/dev/null
cp $(ls /tmp/prof3/*/*kernel_trace.csv | head -1) /tmp/cfg3_trace.csv 2>/dev/null
python tools/trace_gaps.py /tmp/cfg3_trace.csv --steps 3 --top 16 > $OUT/cfg3_trace_gaps.txt 2>&1; head -40 $OUT/cfg3_trace_gaps.txt | cut -c1-200
