#!/bin/bash
# GPU call 15 (round 6): why conv4_2 / conv4_3 take 450-480 us in the step and under the counters but 380 us in the back-to-back sweep:
# kernel-trace durations (no counters) of the replay, 5 launches per layer, auto schedule against whole tiles; then the tower weight
# gradient with 32 channels per wave (two blocks per CU) against 64
set -u
OUT=gpurun_out/r06_c15; mkdir -p $OUT; R=$PWD
LIB=$R/neural-motifs_amd/csrc/libmotifs_hip.so
cd /tmp && export TMPDIR=/tmp
for v in auto whole; do
  rm -rf /tmp/tr; a=""; [ $v = whole ] && a="1"
  timeout 120 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -- $R/tools/_bin/pl_check $LIB --conv-replay 5 $a > /dev/null 2>&1
  T=$(ls /tmp/tr/*/*kernel_trace.csv | head -1)
  python - "$T" $v <<'PY'
import csv, sys
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(sys.argv[1])) if 'conv3x3_ring' in r['Kernel_Name']]
rows.sort()
names = ['conv1_2','conv2_1','conv2_2','conv3_1','conv3_2','conv3_3','conv4_1','conv4_2','conv4_3','conv5_1','conv5_2','conv5_3']
print(sys.argv[2], len(rows))
for i, n in enumerate(names):
    d = [(b - a) / 1e3 for a, b, _ in rows[5 * i:5 * i + 5]]
    print('  %-8s' % n, ' '.join('%7.1f' % x for x in d))
PY
done
cd $R
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "tower" > $OUT/tests_tower.log 2>&1; grep -E "passed|failed|rror" $OUT/tests_tower.log | tail -3 | cut -c1-200
show() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], round(d['value'],1), d['unit'], 'ms', round(d['ms_per_step'],2), 'p50', d.get('ms_per_step_p50'), 'cal', round(d['calibration']['plane_gemm_4096_tflops']),
          'seg', {k: round(v, 2) for k, v in d['main_stream_segments'].items() if k.endswith('_ms')})
except Exception as e:
    print(sys.argv[1], 'unreadable', e)
PY
}
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline"
timeout 200 $B > $OUT/bench_new.json 2> $OUT/bench_new.err; show $OUT/bench_new.json
MH_TOWER_WGRAD_NB=2 timeout 200 $B > $OUT/bench_nb2.json 2> /dev/null; show $OUT/bench_nb2.json
timeout 200 $B > $OUT/bench_new_b.json 2> /dev/null; show $OUT/bench_new_b.json
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_cfg2 -- python $R/bench.py --steps 8 --warmup 4 --no-cpu-baseline --h2d-steps 0 --meter-every 1000 > $R/$OUT/prof_cfg2.log 2>&1 )
cp $(ls /tmp/prof_cfg2/*/*kernel_stats.csv | head -1) $OUT/kernel_stats_cfg2.csv 2>/dev/null
grep -E "tower|image_absmax|bn_finalize" $OUT/kernel_stats_cfg2.csv | cut -d, -f1-4 | cut -c1-200
