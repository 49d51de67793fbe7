#!/bin/bash
# GPU call 3 (round 4): table-free ring conv: accuracy (all shapes, both epilogues) + shape x split-K sweep per trunk layer
set -u
OUT=gpurun_out/r04_c3; mkdir -p $OUT
( timeout 300 tools/_bin/pl_check neural-motifs_amd/csrc/libmotifs_hip.so --conv ) > $OUT/ring_conv_check.jsonl 2>&1
tail -1 $OUT/ring_conv_check.jsonl; grep '"ok": false' $OUT/ring_conv_check.jsonl | cut -c1-300 | head -20; grep -c '"ok": true' $OUT/ring_conv_check.jsonl
( timeout 400 tools/_bin/pl_check neural-motifs_amd/csrc/libmotifs_hip.so --conv-sweep --quick ) > $OUT/ring_conv_sweep.jsonl 2>&1
python - <<'PY'
import json, collections
best = collections.OrderedDict()
for l in open('gpurun_out/r04_c3/ring_conv_sweep.jsonl'):
    try: d = json.loads(l)
    except Exception: print(l[:200]); continue
    k = (d['case'], d['shape'])
    for key, tf in (('fp32', d['tflops']), ('img', d['tflops_image_out'])):
        kk = k + (key,)
        if kk not in best or tf > best[kk][0]: best[kk] = (tf, d['splitk'])
last = None
for k, v in best.items():
    if k[0] != last: print(); last = k[0]; print(k[0], end=': ')
    print('s%d/%s %.0f(k%d)' % (k[1], k[2], v[0], v[1]), end='  ')
print()
PY
