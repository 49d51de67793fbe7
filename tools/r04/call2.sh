#!/bin/bash
# GPU call 2 (round 4): ring conv accuracy (all shapes, both epilogues), shape x split-K sweep per trunk layer, bench with quiet GC
set -u
OUT=gpurun_out/r04_c2; mkdir -p $OUT
( timeout 300 tools/_bin/pl_check neural-motifs_amd/csrc/libmotifs_hip.so --conv ) > $OUT/ring_conv_check.jsonl 2>&1
tail -1 $OUT/ring_conv_check.jsonl; grep '"ok": false' $OUT/ring_conv_check.jsonl | cut -c1-300 | head -20; grep -c '"ok": true' $OUT/ring_conv_check.jsonl
( timeout 400 tools/_bin/pl_check neural-motifs_amd/csrc/libmotifs_hip.so --conv-sweep --quick ) > $OUT/ring_conv_sweep.jsonl 2>&1
python - <<'PY'
import json, collections
best = collections.OrderedDict()
for l in open('gpurun_out/r04_c2/ring_conv_sweep.jsonl'):
    try: d = json.loads(l)
    except Exception: print(l[:200]); continue
    k = (d['case'], d['shape'])
    for key, tf in (('fp32', d['tflops']), ('img', d['tflops_image_out'])):
        kk = k + (key,)
        if kk not in best or tf > best[kk][0]: best[kk] = (tf, d['splitk'])
for k, v in best.items(): print(k, v)
PY
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; tail -2 $OUT/bench.err | cut -c1-300
python -c "
import json; d=json.loads(open('$OUT/bench.json').read().strip().split('\n')[-1]); print('bench', round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms; p50/p90/max', d['ms_per_step_p50'], d['ms_per_step_p90'], d['ms_per_step_max'], '; h2d', d['h2d_inclusive'] and round(d['h2d_inclusive']['ms_per_step'],2), '; calib', round(d['calibration']['plane_gemm_4096_tflops'],1), '; conv', round(d['roofline']['achieved'],1), 'gemm', round(d['roofline_gemm']['achieved'],1), round(d['roofline_gemm']['ms_per_step'],2)); print(d['step_ms']['gpu_per_step'])"
