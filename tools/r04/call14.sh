#!/bin/bash
# GPU call 14 (round 4): whole -m gpu suite, the recorded bench line (with roofline.traffic + CPU baseline), kernel trace of the cfg3 step
set -u
OUT=gpurun_out/r04_c14; mkdir -p $OUT; R=$PWD
timeout 1300 python -m pytest tests/ -x -q -m gpu > $OUT/gpu_tests.log 2>&1; tail -3 $OUT/gpu_tests.log | cut -c1-300
timeout 400 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; tail -1 $OUT/bench.json | cut -c1-200
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof3 -- python $R/bench.py --config cfg3 --steps 6 --warmup 3 --no-cpu-baseline > $R/$OUT/prof_cfg3.log 2>&1 )
cp $(ls /tmp/prof3/*/*kernel_stats.csv | head -1) $OUT/cfg3_kernel_stats.csv 2>/dev/null
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r04_c14/cfg3_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows); steps=9
print('cfg3 kernel time per step %.2f ms' % (tot/steps/1e6))
for r in rows[:28]: print('%-86s %6.1f/step %7.3f ms/step %8.1f us' % (r['Name'][:86], int(r['Calls'])/steps, float(r['TotalDurationNs'])/steps/1e6, float(r['AverageNs'])/1e3))
PY
tail -1 $OUT/prof_cfg3.log | cut -c1-200
