#!/bin/bash
# GPU call 27: same-box A/B of the plane conv with (new) and without (old) the per-k-tile offset table
set -u
OUT=gpurun_out/r03_c27; mkdir -p $OUT
V=$PWD/neural-motifs_amd/csrc/_variants/oldconv/libmotifs_hip.so
for rep in 1 2; do for lib in old new; do
  if [ $lib = old ]; then export MOTIFS_HIP_LIB=$V; else unset MOTIFS_HIP_LIB; fi
  timeout 60 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_${lib}_$rep.json
  python -c "
import json; d=json.loads(open('$OUT/bench_${lib}_$rep.json').read()); t=d['roofline']['trunk_only']; print('$lib rep $rep:', round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms; calib', round(d['calibration']['plane_gemm_4096_tflops'],1), '; trunk', round(t['tflops'],1), 'TF', round(t['ms_per_step'],2), 'ms; gemm', round(d['roofline_gemm']['ms_per_step'],2), 'ms')"
done; done
