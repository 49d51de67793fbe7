#!/bin/bash
# GPU call 19 (round 6): cfg1 (PredCls evaluation, one image per step) is slower than on round 5's tree (265 / 222 against 286 img/s, r06_c18):
# kernel statistics of both trees on the same box
set -u
OUT=gpurun_out/r06_c19; mkdir -p $OUT; R=$PWD
cd /tmp && export TMPDIR=/tmp
for t in new r05; do
  D=$R; [ $t = r05 ] && D=$R/_ab/r05
  rm -rf /tmp/p_$t; ( cd $D && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$t -- python bench.py --config cfg1 --steps 12 --warmup 4 > $R/$OUT/prof_$t.log 2>&1 )
  cp $(ls /tmp/p_$t/*/*kernel_stats.csv | head -1) $R/$OUT/kernel_stats_cfg1_$t.csv 2>/dev/null
  tail -1 $R/$OUT/prof_$t.log | cut -c1-200
done
cd $R
python - <<'PY'
import csv
def load(f):
    d = {}
    for r in csv.DictReader(open(f)):
        d[r['Name'][:70]] = (int(r['Calls']), float(r['TotalDurationNs']) / 1e6)
    return d
a, b = load('gpurun_out/r06_c19/kernel_stats_cfg1_new.csv'), load('gpurun_out/r06_c19/kernel_stats_cfg1_r05.csv')
print('total ms new %.1f (%d launches)   r05 %.1f (%d launches)' % (sum(v[1] for v in a.values()), sum(v[0] for v in a.values()), sum(v[1] for v in b.values()), sum(v[0] for v in b.values())))
keys = sorted(set(a) | set(b), key=lambda k: -abs(a.get(k, (0, 0))[1] - b.get(k, (0, 0))[1]))
for k in keys[:28]:
    x, y = a.get(k, (0, 0.0)), b.get(k, (0, 0.0))
    print('%-72s new %4d %8.2f   r05 %4d %8.2f   diff %+7.2f' % (k, x[0], x[1], y[0], y[1], x[1] - y[1]))
PY
