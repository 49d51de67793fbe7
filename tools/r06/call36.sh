#!/bin/bash
# GPU call 36 (round 6): records of the final tree -- every matrix product of a cfg2 step (--gemm-shapes), kernel statistics of the cfg4 step
set -u
OUT=gpurun_out/r06_c36; mkdir -p $OUT; R=$PWD
timeout 300 python bench.py --steps 10 --warmup 5 --no-cpu-baseline --gemm-shapes $OUT/gemm_shapes.jsonl > $OUT/bench.json 2> $OUT/bench.err; tail -c 300 $OUT/bench.json | cut -c1-200
python - <<'PY'
import json
rows=[json.loads(l) for l in open('gpurun_out/r06_c36/gemm_shapes.jsonl')]
for r in rows:
    if r['gflop'] > 20: print(r['binding'], r['M'], r['N'], r['K'], r['us'], 'us', round(r['gflop']/r['us']*1e-3,1), 'TF/s', round(r['gflop']/r['us']*1e-3/833.3,3), 'stream', r['stream'])
PY
( cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_cfg4 -- python $R/bench.py --config cfg4 --steps 6 --warmup 3 > $R/$OUT/prof_cfg4.log 2>&1 )
cp $(ls /tmp/prof_cfg4/*/*kernel_stats.csv | head -1) $OUT/kernel_stats_cfg4.csv 2>/dev/null; head -6 $OUT/kernel_stats_cfg4.csv | cut -c1-150
