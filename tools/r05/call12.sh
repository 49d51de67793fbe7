#!/bin/bash
# GPU call 12 (round 5): per-launch kernel trace of the cfg4 (ResNet-101) step on the current tree, and the per-layer table of the
# VGG trunk's plane convs (pl_check --conv-sweep --quick: every block-tile shape x K split per layer; the planner's row = splitk 0)
set -u
OUT=gpurun_out/r05_c12; mkdir -p $OUT; R=$PWD
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_cfg4 -- python $R/bench.py --config cfg4 --steps 3 --warmup 2 --no-cpu-baseline --meter-every 100 > $R/$OUT/prof_cfg4.log 2>&1 )
cp $(ls /tmp/prof_cfg4/*/*kernel_stats.csv | head -1) $OUT/kernel_stats_cfg4.csv 2>/dev/null
python - <<'PY'
import csv, glob, gzip
f = glob.glob('/tmp/prof_cfg4/*/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
t0 = int(rows[0]['Start_Timestamp'])
with gzip.open('gpurun_out/r05_c12/kernel_trace_cfg4.csv.gz', 'wt') as o:
    o.write('start_us,dur_us,stream,grid,wg,name\n')
    for r in rows:
        o.write('%.1f,%.1f,%s,%s,%s,"%s"\n' % ((int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3,
                                              r.get('Stream_Id', r.get('Queue_Id', '')), r.get('Grid_Size', r.get('Grid_Size_X', '')), r.get('Workgroup_Size', r.get('Workgroup_Size_X', '')), r['Kernel_Name'][:90]))
print(len(rows), 'launches')
PY
tail -1 $OUT/prof_cfg4.log | cut -c1-200
( timeout 300 tools/_bin/pl_check neural-motifs_amd/csrc/libmotifs_hip.so --conv-sweep --quick ) > $OUT/conv_sweep.jsonl 2>&1
grep -c "conv sweep" $OUT/conv_sweep.jsonl; grep '"splitk": 0' $OUT/conv_sweep.jsonl | cut -c1-200 | head -40
