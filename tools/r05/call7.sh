#!/bin/bash
# GPU call 7 (round 5): the whole -m gpu suite + smoke on the final tree; the recorded cfg2 line (with the CPU baseline); rocprofv3
# per-kernel statistics / timeline of the same tree; secondary rows
set -u
OUT=gpurun_out/r05_c7; mkdir -p $OUT; R=$PWD
timeout 1500 python -m pytest tests/ -x -q -m gpu > $OUT/gpu_tests.log 2>&1; grep -E "passed|failed" $OUT/gpu_tests.log | tail -2 | cut -c1-300
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | cut -c1-200
timeout 400 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_c7/bench.json').read().strip().splitlines()[-1])
g=d['step_ms']['gpu_per_step']
print('cfg2', round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms p50', d['ms_per_step_p50'], 'unmetered', round(d['unmetered']['value'],1), 'h2d', round(d['h2d_inclusive']['value'],1),
      'dominant', d['roofline']['dominant_class'], round(d['roofline']['frac'],3), 'gemm', round(d['roofline_gemm']['frac'],3), 'conv', round(d['roofline_conv']['frac'],3),
      'cpu', d.get('cpu_baseline',{}).get('value'), 'cal', round(d['calibration']['plane_gemm_4096_tflops']), 'first12', round(sum(g[:12])/12,2), 'last8', round(sum(g[12:])/8,2))
PY
( cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_cfg2 -- python $R/bench.py --steps 8 --warmup 4 --no-cpu-baseline --h2d-steps 0 --meter-every 1000 > $R/$OUT/prof_cfg2.log 2>&1 )
cp $(ls /tmp/prof_cfg2/*/*kernel_stats.csv | head -1) $OUT/kernel_stats_cfg2.csv 2>/dev/null
T=$(ls /tmp/prof_cfg2/*/*kernel_trace.csv | head -1)
python tools/trace_gaps.py $T --steps 3 --top 8 > $OUT/trace_gaps_cfg2.txt 2>&1; head -4 $OUT/trace_gaps_cfg2.txt | cut -c1-200
python tools/r04/step_timeline.py $T > $OUT/step_timeline.txt 2>&1; head -4 $OUT/step_timeline.txt | cut -c1-200
row() { python -c "
import json; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', round(d['value'],1), d['unit'], round(d['ms_per_step'],2), 'ms', d['config'].get('rows'), d['config'].get('dets'))" 2>&1 | cut -c1-200; }
for c in cfg3 cfg1 cfg4 cfg5 recipe; do timeout 300 python bench.py --config $c --steps 10 --warmup 3 --no-cpu-baseline 2>$OUT/bench_$c.err | tail -1 > $OUT/bench_$c.json; row $OUT/bench_$c.json $c; done
