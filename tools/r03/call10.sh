#!/bin/bash
# GPU call 10: op-level diagnosis of the two-stream race; conv schedule sweep; trunk engine A/B on the bench step
set -u
OUT=gpurun_out/r03_c10; mkdir -p $OUT
timeout 300 python tools/r03/diag4_ops.py 2>&1 | grep -E "^img|Error|error" > $OUT/diag4.log; cat $OUT/diag4.log | cut -c1-1500
MOTIFS_H2D=pageable timeout 240 python tools/r03/diag3_streams.py h2d_pageable 2>&1 | grep VARIANT | tee -a $OUT/diag3.log
( timeout 300 tools/_bin/pl_check neural-motifs_amd/csrc/libmotifs_hip.so --conv-sweep ) > $OUT/conv_sweep.jsonl 2>&1
python - <<'PY'
import json
for l in open('gpurun_out/r03_c10/conv_sweep.jsonl'):
    try: d = json.loads(l)
    except Exception: continue
    print(d.get('case'), 'shape', d.get('shape'), 'sk', d.get('splitk'), 'fp32 %.1f' % d.get('tflops', 0), 'img %.1f' % d.get('tflops_image_out', 0))
PY
for t in planes v2; do
  MOTIFS_TRUNK=$t timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 > $OUT/bench_$t.log
  python -c "
import json,sys
d=json.loads(open('$OUT/bench_$t.log').read()); print('$t', d['value'], d['ms_per_step'], {k:(round(v['tflops'],1), round(v['ms_per_step'],2)) for k,v in d.get('kernels',{}).items()})"
done
