#!/bin/bash
# GPU call 11: sequence-level diagnosis of the two-stream race; bench with the new conv schedule (+ calibration); cfg3 host profile
set -u
OUT=gpurun_out/r03_c11; mkdir -p $OUT
timeout 400 python tools/r03/diag5_seq.py 2>&1 | grep -E "^img|^   context|Error|error|sequence" > $OUT/diag5.log; cut -c1-2500 $OUT/diag5.log
( timeout 200 tools/_bin/pl_check neural-motifs_amd/csrc/libmotifs_hip.so --conv ) > $OUT/pl_conv_check.jsonl 2>&1; tail -1 $OUT/pl_conv_check.jsonl
for t in planes v2; do
  MOTIFS_TRUNK=$t timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 > $OUT/bench_$t.log
  python -c "
import json,sys
d=json.loads(open('$OUT/bench_$t.log').read()); print('$t', round(d['value'],1), round(d['ms_per_step'],2), 'calib', round(d['calibration']['plane_gemm_4096_tflops'],1), 'conv', round(d['roofline']['achieved'],1), d['roofline']['trunk_only'], 'gemm', round(d['roofline_gemm']['achieved'],1))"
done
timeout 300 python bench.py --config cfg3 --steps 10 --warmup 3 --host-profile > $OUT/bench_cfg3.log 2> $OUT/bench_cfg3_host.txt; tail -1 $OUT/bench_cfg3.log | cut -c1-400; grep -m1 "host enqueue" $OUT/bench_cfg3_host.txt; head -45 $OUT/bench_cfg3_host.txt | tail -38
