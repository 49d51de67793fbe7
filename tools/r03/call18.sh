#!/bin/bash
set -u
OUT=gpurun_out/r03_c18; mkdir -p $OUT
V=$PWD/neural-motifs_amd/csrc/_variants
MOTIFS_HIP_LIB=$V/pk/libmotifs_hip.so timeout 400 python tools/r03/diag9_pairs.py packed 2>&1 | grep -E "^PAIR|Error|error" > $OUT/diag9.log
timeout 400 python tools/r03/diag9_pairs.py scalar 2>&1 | grep -E "^PAIR|Error|error" >> $OUT/diag9.log
grep -v "wrong  0 of" $OUT/diag9.log | cut -c1-230; echo "rows: $(wc -l < $OUT/diag9.log), clean: $(grep -c 'wrong  0 of' $OUT/diag9.log)"
timeout 240 python tools/r03/diag3_streams.py no_packed_f32 2>&1 | grep VARIANT | tee $OUT/diag3.log
timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_sgdet.py -q 2>&1 | tail -5 | cut -c1-300
for lib in pk default; do
  if [ $lib = pk ]; then export MOTIFS_HIP_LIB=$V/pk/libmotifs_hip.so; else unset MOTIFS_HIP_LIB; fi
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 > $OUT/bench_$lib.log
  python -c "
import json,sys
d=json.loads(open('$OUT/bench_$lib.log').read()); print('$lib', round(d['value'],1), round(d['ms_per_step'],2), 'calib', round(d['calibration']['plane_gemm_4096_tflops'],1), 'conv', round(d['roofline']['achieved'],1), 'trunk', round(d['roofline']['trunk_only']['tflops'],1), round(d['roofline']['trunk_only']['ms_per_step'],2), 'gemm', round(d['roofline_gemm']['achieved'],1), round(d['roofline_gemm']['ms_per_step'],2))"
done
