#!/bin/bash
# HBM-side traffic of the dominant kernel during bench.py (run on the GPU box): one --pmc pass, FETCH_SIZE / WRITE_SIZE
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
timeout 400 rocprofv3 --pmc FETCH_SIZE WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmct -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /tmp/pmct.log 2>&1
f=$(ls /tmp/pmct/*/*counter_collection.csv 2>/dev/null | head -1)
if [ -z "$f" ]; then tail -20 /tmp/pmct.log; exit 1; fi
python - "$f" "$OUT/r01_pmc_bench_traffic.csv" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r['Kernel_Name'].split('(')[0][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
with open(sys.argv[2], 'w') as fo:
    fo.write('# rocprofv3 --pmc FETCH_SIZE WRITE_SIZE over bench.py --steps 2 --warmup 1 (units as reported: KB); per kernel: launches, mean per launch\n')
    fo.write('kernel,launches,FETCH_SIZE_mean,WRITE_SIZE_mean\n')
    rows = []
    for k, d in acc.items():
        n = len(d.get('FETCH_SIZE', []))
        if n: rows.append((sum(d['FETCH_SIZE']), k, n, sum(d['FETCH_SIZE']) / n, sum(d.get('WRITE_SIZE', [0])) / max(1, len(d.get('WRITE_SIZE', [0])))))
    for _, k, n, f, w in sorted(rows, reverse=True)[:40]:
        fo.write('%s,%d,%.1f,%.1f\n' % (k, n, f, w))
print(open(sys.argv[2]).read()[:2500])
PY
