#!/bin/bash
# GPU call 11 (round 5): 3x3 convs over many small maps on the ring engine (mask tower; ResNet layer4): unit test against float64,
# the tower / cfg2 / cfg4 parity tests on the new path, cfg2 and cfg4 bench A/B (MOTIFS_CONV3X3_MAPS=inloop = the round-2 kernels)
set -u
OUT=gpurun_out/r05_c11; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "small_maps or tower or conv3x3" > $OUT/ops.log 2>&1; grep -E "passed|failed|rror" $OUT/ops.log | tail -3 | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "cfg2 or cfg4" > $OUT/cfg.log 2>&1; grep -E "passed|failed|rror" $OUT/cfg.log | tail -3 | cut -c1-300
row() { python -c "
import json; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', round(d['value'],1), d['unit'], round(d['ms_per_step'],2), 'ms', d.get('unmetered',{}).get('value') if isinstance(d.get('unmetered'),dict) else '')" 2>&1 | cut -c1-300; }
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline"
timeout 200 $B > $OUT/bench_planes.json 2>$OUT/bench_planes.err; row $OUT/bench_planes.json cfg2_planes
MOTIFS_CONV3X3_MAPS=inloop timeout 200 $B > $OUT/bench_inloop.json 2>/dev/null; row $OUT/bench_inloop.json cfg2_inloop
timeout 200 $B > $OUT/bench_planes_b.json 2>/dev/null; row $OUT/bench_planes_b.json cfg2_planes_b
timeout 300 python bench.py --config cfg4 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_cfg4.json; row $OUT/bench_cfg4.json cfg4_planes
MOTIFS_CONV3X3_MAPS=inloop timeout 300 python bench.py --config cfg4 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_cfg4_inloop.json; row $OUT/bench_cfg4_inloop.json cfg4_inloop
tail -n 3 $OUT/bench_planes.err | cut -c1-200
