#!/bin/bash
# GPU call 1 (round 4): ring GEMM accuracy + speed, vendor f16 GEMM ceiling, step-time variance probe, a bench line
set -u
OUT=gpurun_out/r04_c1; mkdir -p $OUT
( timeout 240 tools/_bin/pl_check neural-motifs_amd/csrc/libmotifs_hip.so --ring ) > $OUT/ring_check.jsonl 2>&1
grep -c '"ok": false' $OUT/ring_check.jsonl; grep "ring accuracy summary" $OUT/ring_check.jsonl; grep "ring speed" $OUT/ring_check.jsonl | python -c "
import sys, json
best = {}
for l in sys.stdin:
    d = json.loads(l); k = (d['case'], d['shape'])
    if k not in best or d['tflops'] > best[k][0]: best[k] = (d['tflops'], d['splitk'])
for k in sorted(best): print(k, best[k])"
timeout 240 python tools/r04/vendor_gemm.py > $OUT/vendor_gemm.jsonl 2> $OUT/vendor_gemm.err; cut -c1-230 $OUT/vendor_gemm.jsonl; tail -3 $OUT/vendor_gemm.err
timeout 300 python tools/r04/variance_probe.py --regions 3 --steps 20 --out $OUT/variance.jsonl > $OUT/variance.log 2> $OUT/variance.err; tail -1 $OUT/variance.log | cut -c1-1200; tail -3 $OUT/variance.err
python - <<'PY'
import json
for l in open('gpurun_out/r04_c1/variance.jsonl'):
    d = json.loads(l)
    if 'arm' in d: print(d['arm'], d['region'], 'wall', d['wall_ms_per_step'], 'gpu p50/p90/max', d['gpu_p50'], d['gpu_p90'], d['gpu_max'], 'host p50/max', d['host_p50'], d['host_max'], 'alloc', d['alloc_events'][:3], 'gc', d['gc_collections'])
PY
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; tail -2 $OUT/bench.err | cut -c1-300
python -c "
import json; d=json.loads(open('$OUT/bench.json').read().strip().split('\n')[-1]); print('bench', round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms; p50/p90/max', d['ms_per_step_p50'], d['ms_per_step_p90'], d['ms_per_step_max'], '; h2d', d['h2d_inclusive'] and round(d['h2d_inclusive']['ms_per_step'],2), '; calib', round(d['calibration']['plane_gemm_4096_tflops'],1), '; conv', round(d['roofline']['achieved'],1), 'gemm', round(d['roofline_gemm']['achieved'],1), round(d['roofline_gemm']['ms_per_step'],2)); print(d['step_ms']['gpu_per_step'])"
