#!/bin/bash
# GPU call 17 (round 5): the FINAL tree -- whole -m gpu suite + smoke, the recorded cfg2 line (with the CPU baseline), rocprofv3 trace of
# the SGDet step with the detector stage two batches ahead, secondary rows
set -u
OUT=gpurun_out/r05_c17; mkdir -p $OUT; R=$PWD
timeout 1500 python -m pytest tests/ -x -q -m gpu > $OUT/gpu_tests.log 2>&1; grep -E "passed|failed" $OUT/gpu_tests.log | tail -2 | cut -c1-300
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | cut -c1-200
timeout 400 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_c17/bench.json').read().strip().splitlines()[-1])
g=d['step_ms']['gpu_per_step']
print('cfg2', round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms p50', d['ms_per_step_p50'], 'unmetered', round(d['unmetered']['value'],1), 'h2d', round(d['h2d_inclusive']['value'],1),
      'dominant', d['roofline']['dominant_class'], round(d['roofline']['frac'],3), 'gemm', round(d['roofline_gemm']['frac'],3), 'conv', round(d['roofline_conv']['frac'],3),
      'cpu', d.get('cpu_baseline',{}).get('value'), 'cal', round(d['calibration']['plane_gemm_4096_tflops']), 'first12', round(sum(g[:12])/12,2), 'last8', round(sum(g[12:])/8,2))
PY
row() { python -c "
import json; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', round(d['value'],1), d['unit'], round(d['ms_per_step'],2), 'ms', d['config'].get('rows'), d['config'].get('dets'), d['config'].get('detector_stage'))" 2>&1 | cut -c1-240; }
for c in cfg3 cfg1 cfg4 cfg5 recipe; do timeout 300 python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline 2>$OUT/bench_$c.err | tail -1 > $OUT/bench_$c.json; row $OUT/bench_$c.json $c; done
MOTIFS_DETECT_AHEAD=0 timeout 300 python bench.py --config cfg3 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_cfg3_inline.json; row $OUT/bench_cfg3_inline.json cfg3_inline
MOTIFS_AHEAD_PRIORITY=-1 timeout 300 python bench.py --config cfg3 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_cfg3_high.json; row $OUT/bench_cfg3_high.json cfg3_high_priority
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_cfg3 -- python $R/bench.py --config cfg3 --steps 8 --warmup 4 --no-cpu-baseline --meter-every 100 > $R/$OUT/prof_cfg3.log 2>&1 )
T=$(ls /tmp/prof_cfg3/*/*kernel_trace.csv | head -1)
cp $(ls /tmp/prof_cfg3/*/*kernel_stats.csv | head -1) $OUT/kernel_stats_cfg3.csv 2>/dev/null
python tools/trace_gaps.py $T --steps 3 --top 8 > $OUT/trace_gaps_cfg3.txt 2>&1; head -8 $OUT/trace_gaps_cfg3.txt | cut -c1-200
