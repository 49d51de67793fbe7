#!/bin/bash
# GPU call 35 (round 6): the final tree -- whole -m gpu suite, smoke(), the flag-less default bench (with the CPU baseline) = the recorded
# line, a 20-step line, kernel statistics / timeline / every launch of a step, secondary configurations
set -u
OUT=gpurun_out/r06_c35; mkdir -p $OUT; R=$PWD
timeout 2400 python -m pytest tests -x -q -m gpu > $OUT/tests_all.log 2>&1; grep -E "passed|failed|rror" $OUT/tests_all.log | tail -5 | cut -c1-300
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log | cut -c1-300
show() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], round(d['value'],1), d['unit'], 'ms', round(d['ms_per_step'],2), 'p50', d.get('ms_per_step_p50'),
          'gemm', round(d['roofline_gemm']['frac'],3), 'imgs', round(d['roofline_gemm']['products_on_images']['frac'],3),
          'conv', round(d['roofline_conv']['frac'],3), 'trunk', round(d['roofline']['frac_trunk_only'],3), 'cal', round(d['calibration']['plane_gemm_4096_tflops']),
          'cpu', (d.get('cpu_baseline') or {}).get('value'), 'unmetered', round((d.get('unmetered') or {}).get('value', 0), 1), 'h2d', round((d.get('h2d_inclusive') or {}).get('value', 0), 1),
          'seg', {k: round(v, 2) for k, v in d['main_stream_segments'].items() if k.endswith('_ms')})
except Exception as e:
    print(sys.argv[1], 'unreadable', e)
PY
}
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; show $OUT/bench_default.json
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_20.json 2> /dev/null; show $OUT/bench_20.json
timeout 300 python bench.py --steps 24 --warmup 8 --no-cpu-baseline > $OUT/bench_24w8.json 2> /dev/null; show $OUT/bench_24w8.json
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_cfg2 -- python $R/bench.py --steps 8 --warmup 4 --no-cpu-baseline --h2d-steps 0 --meter-every 1000 > $R/$OUT/prof_cfg2.log 2>&1 )
cp $(ls /tmp/prof_cfg2/*/*kernel_stats.csv | head -1) $OUT/kernel_stats_cfg2.csv 2>/dev/null
T=$(ls /tmp/prof_cfg2/*/*kernel_trace.csv | head -1)
python tools/trace_gaps.py $T --steps 3 --top 8 > $OUT/trace_gaps_cfg2.txt 2>&1; head -3 $OUT/trace_gaps_cfg2.txt | cut -c1-200
python tools/r04/step_timeline.py $T > $OUT/step_timeline.txt 2>&1; head -3 $OUT/step_timeline.txt | cut -c1-200
python tools/r04/step_timeline.py $T --all > $OUT/step_launches.txt 2>&1
for c in cfg1 cfg3 cfg4 cfg5 recipe; do timeout 400 python bench.py --config $c --steps 10 --warmup 3 2>/dev/null | tail -1 > $OUT/bench_$c.json; python -c "
import json; d=json.loads(open('$OUT/bench_$c.json').read()); print('$c', round(d['value'],1), d['unit'], round(d['ms_per_step'],2), 'ms')"; done
