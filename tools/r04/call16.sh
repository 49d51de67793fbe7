#!/bin/bash
# GPU call 16 (round 4): final validation of the tree: pl_check accuracy (GEMM + conv, all shapes), whole -m gpu suite, smoke, the recorded bench
# line with the CPU baseline, secondary configurations
set -u
OUT=gpurun_out/r04_c16; mkdir -p $OUT
( timeout 200 tools/_bin/pl_check neural-motifs_amd/csrc/libmotifs_hip.so --conv ) > $OUT/conv_check.jsonl 2>&1; tail -1 $OUT/conv_check.jsonl
( timeout 100 tools/_bin/pl_check neural-motifs_amd/csrc/libmotifs_hip.so --ring --accuracy ) > $OUT/gemm_check.jsonl 2>&1; tail -1 $OUT/gemm_check.jsonl
timeout 1300 python -m pytest tests/ -x -q -m gpu > $OUT/gpu_tests.log 2>&1; tail -2 $OUT/gpu_tests.log | cut -c1-300
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | cut -c1-200
timeout 400 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; tail -1 $OUT/bench.json | cut -c1-250
for c in cfg1 cfg3 cfg4 cfg5 recipe; do timeout 300 python bench.py --config $c --steps 10 --warmup 3 2>/dev/null | tail -1 > $OUT/bench_$c.json; python -c "
import json; d=json.loads(open('$OUT/bench_$c.json').read()); print('$c', round(d['value'],1), d['unit'], round(d['ms_per_step'],2), 'ms')"; done
