#!/bin/bash
# NOT YET RUN (written after round 3's GPU budget was spent; DESIGN.md section 7 item 00 is its purpose).
# Why does one binary on one box give 258..299 img/s?  (a) distribution of the bench value with two streams and with one,
# (b) a kernel trace of every two-stream run: slow and fast runs can then be compared kernel by kernel
#     (tools/trace_gaps.py per run; the persistent LSTM launches' start / duration against the big GEMMs).
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/r03/variance_probe.sh'
set -u
R=$PWD; OUT=$R/gpurun_out/variance; mkdir -p $OUT
for rep in 1 2 3 4; do for ov in 1 0; do
  MOTIFS_OVERLAP=$ov timeout 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_ov${ov}_$rep.json
  python -c "
import json; d=json.loads(open('$OUT/bench_ov${ov}_$rep.json').read()); print('overlap $ov rep $rep:', round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms; calib', round(d['calibration']['plane_gemm_4096_tflops'],1), '; gemm', round(d['roofline_gemm']['ms_per_step'],2), 'ms; lstm fwd/bwd us', round(d['hbm_kernels']['lstm_fwd']['us_per_call']), round(d['hbm_kernels']['lstm_bwd']['us_per_call']))"
done; done
for rep in 1 2 3; do
  ( cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/var_$rep -- python $R/bench.py --steps 8 --warmup 4 --no-cpu-baseline > $OUT/traced_$rep.log 2>&1 )
  python tools/trace_gaps.py $(ls /tmp/var_$rep/*/*kernel_trace.csv | head -1) --steps 3 --top 8 > $OUT/trace_gaps_$rep.txt 2>&1
  tail -1 $OUT/traced_$rep.log | cut -c1-120; head -3 $OUT/trace_gaps_$rep.txt | cut -c1-160
done
