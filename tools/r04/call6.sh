#!/bin/bash
# GPU call 6 (round 4): transposed-accumulator epilogues (ring conv + ring GEMM): accuracy, then sweeps
set -u
OUT=gpurun_out/r04_c6; mkdir -p $OUT
( timeout 300 tools/_bin/pl_check neural-motifs_amd/csrc/libmotifs_hip.so --conv ) > $OUT/ring_conv_check.jsonl 2>&1
tail -1 $OUT/ring_conv_check.jsonl; grep '"ok": false' $OUT/ring_conv_check.jsonl | cut -c1-330 | head -12; grep -c '"ok": true' $OUT/ring_conv_check.jsonl
( timeout 200 tools/_bin/pl_check neural-motifs_amd/csrc/libmotifs_hip.so --ring --accuracy ) > $OUT/ring_gemm_check.jsonl 2>&1
tail -1 $OUT/ring_gemm_check.jsonl; grep '"ok": false' $OUT/ring_gemm_check.jsonl | cut -c1-300 | head -8
( timeout 400 tools/_bin/pl_check neural-motifs_amd/csrc/libmotifs_hip.so --conv-sweep --quick ) > $OUT/ring_conv_sweep.jsonl 2>&1
python - <<'PY'
import json, collections
best = collections.OrderedDict()
for l in open('gpurun_out/r04_c6/ring_conv_sweep.jsonl'):
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
( timeout 200 tools/_bin/pl_check neural-motifs_amd/csrc/libmotifs_hip.so --ring ) > $OUT/ring_gemm.jsonl 2>&1
grep "ring speed" $OUT/ring_gemm.jsonl | python -c "
import sys, json
best = {}
for l in sys.stdin:
    d = json.loads(l); k = (d['case'], d['shape'])
    if k not in best or d['tflops'] > best[k][0]: best[k] = (d['tflops'], d['splitk'])
for k in sorted(best): print(k, best[k])"
