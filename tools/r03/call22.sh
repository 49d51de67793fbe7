#!/bin/bash
# GPU call 22: final measurements of the round on the shipped tree
set -u
OUT=gpurun_out/r03_c22; mkdir -p $OUT; R=$PWD
timeout 600 python -m pytest tests/test_gpu_sgdet.py -x -q -s -k "cfg5" > $OUT/cfg5.log 2>&1; grep -E "cfg5|passed|failed" $OUT/cfg5.log | tail -8 | cut -c1-220
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | cut -c1-200
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -1 $OUT/bench.json | cut -c1-400
for c in cfg1 cfg3 cfg4 cfg5 recipe; do timeout 300 python bench.py --config $c --steps 10 --warmup 3 2>/dev/null | tail -1 > $OUT/bench_$c.json; python -c "
import json; d=json.loads(open('$OUT/bench_$c.json').read()); print('$c', round(d['value'],1), d['unit'], round(d['ms_per_step'],2), 'ms')"; done
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $R/bench.py --steps 8 --warmup 4 --no-cpu-baseline > $R/$OUT/prof_bench.log 2>&1 )
cp $(ls /tmp/prof/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv 2>/dev/null
cp $(ls /tmp/prof/*/*kernel_trace.csv | head -1) /tmp/kernel_trace.csv 2>/dev/null
python tools/trace_gaps.py /tmp/kernel_trace.csv --steps 3 --top 14 > $OUT/trace_gaps.txt 2>&1; head -6 $OUT/trace_gaps.txt | cut -c1-200
tail -1 $OUT/prof_bench.log | cut -c1-200
