#!/bin/bash
# GPU call 3 (round 5): the whole -m gpu suite on the tree with the small-product engine, the host-side packing order, the
# time-bounded granule sweep, the segmented reducer and the SGDet context-first order; smoke; secondary bench rows (cfg3 A/B of the
# context-first order, cfg1, cfg4, cfg5, recipe)
set -u
OUT=gpurun_out/r05_c3; mkdir -p $OUT; R=$PWD
timeout 1500 python -m pytest tests/ -x -q -m gpu > $OUT/gpu_tests.log 2>&1; tail -3 $OUT/gpu_tests.log | cut -c1-300
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | cut -c1-200
row() { python -c "
import json; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', round(d['value'],1), d['unit'], round(d['ms_per_step'],2), 'ms', d['config'].get('rows'), d['config'].get('dets'), 'gemm frac', round(d['kernels']['gemm']['tflops']/d['roofline']['peak'],3) if 'kernels' in d else '')" 2>&1 | cut -c1-200; }
for c in cfg3 cfg1 cfg4 cfg5 recipe; do timeout 300 python bench.py --config $c --steps 10 --warmup 3 --no-cpu-baseline 2>$OUT/bench_$c.err | tail -1 > $OUT/bench_$c.json; row $OUT/bench_$c.json $c; done
MOTIFS_SGDET_CONTEXT_FIRST=0 timeout 300 python bench.py --config cfg3 --steps 10 --warmup 3 2>/dev/null | tail -1 > $OUT/bench_cfg3_old_order.json; row $OUT/bench_cfg3_old_order.json cfg3_old_order
timeout 300 python bench.py --config cfg3 --steps 10 --warmup 3 2>/dev/null | tail -1 > $OUT/bench_cfg3_b.json; row $OUT/bench_cfg3_b.json cfg3_b
