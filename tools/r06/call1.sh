#!/bin/bash
# GPU call 1 (round 6): (a) per-image maxima: atomicMax only when it can raise the word (csrc/pl_tile.h: may_raise) -- the trunk's
# per-layer table against the MH_ATOMIC_ALWAYS=1 build of the same tree; (b) XCD-banded tile order of the plane GEMMs (gemm_item):
# GEMM tests on the new order, speed A/B per product (pl_check --order-ab), fabric traffic A/B (rocprofv3 --pmc FETCH_SIZE /
# WRITE_SIZE passes over tools/_bin/gemm_traffic), then the cfg2 bench line A/B.
set -u
OUT=gpurun_out/r06_c1; mkdir -p $OUT; R=$PWD
LIB=neural-motifs_amd/csrc/libmotifs_hip.so; OLD=neural-motifs_amd/csrc/_variants/atomics/libmotifs_hip.so
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "gemm or plane or linear or small_product" > $OUT/ops_tests.log 2>&1; grep -E "passed|failed|rror" $OUT/ops_tests.log | tail -3 | cut -c1-300
( timeout 200 tools/_bin/pl_check $LIB --ring --accuracy ) > $OUT/ring_accuracy.jsonl 2>&1; tail -1 $OUT/ring_accuracy.jsonl | cut -c1-200
( timeout 300 tools/_bin/pl_check $LIB --conv-sweep --quick ) > $OUT/conv_sweep_new.jsonl 2>&1
( timeout 300 tools/_bin/pl_check $OLD --conv-sweep --quick ) > $OUT/conv_sweep_atomics_always.jsonl 2>&1
( timeout 300 tools/_bin/pl_check $LIB --conv-sweep --quick ) > $OUT/conv_sweep_new_b.jsonl 2>&1
for f in new atomics_always new_b; do echo "== $f"; grep -E '"splitk": (0|1),' $OUT/conv_sweep_$f.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    if d['splitk'] == 0: print(d['case'], d['shape'], d['ms'], d['tflops'], d['ms_image_out'], d['tflops_image_out'])
"; done
( timeout 400 tools/_bin/pl_check $LIB --order-ab ) > $OUT/order_ab.jsonl 2>&1
python - <<'PY'
import json
rows = [json.loads(l) for l in open('gpurun_out/r06_c1/order_ab.jsonl') if l.startswith('{"check": "order ab"')]
for r in rows:
    print(r['case'], 'order', r['order'], 'shape', r['shape'], 'sk', r['splitk'], r['ms'], r['tflops'])
PY
bash tools/traffic_run.sh gemm old=MH_GEMM_ORDER=0 new=MH_GEMM_ORDER=1 > $OUT/traffic_run.log 2>&1
cp gpurun_out/traffic/gemm.* $OUT/ 2>/dev/null; tail -2 $OUT/traffic_run.log | cut -c1-200
show() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], round(d['value'],1), d['unit'], 'ms', round(d['ms_per_step'],2), 'p50', d.get('ms_per_step_p50'),
          'gemm', round(d['roofline_gemm']['frac'],3), round(d['roofline_gemm']['ms_per_step'],2), 'imgs', round(d['roofline_gemm']['products_on_images']['frac'],3),
          'conv', round(d['roofline_conv']['frac'],3), 'trunk', round(d['roofline']['frac_trunk_only'],3), 'cal', round(d['calibration']['plane_gemm_4096_tflops']))
except Exception as e:
    print(sys.argv[1], 'unreadable', e)
PY
}
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline"
timeout 200 $B > $OUT/bench_new.json 2> $OUT/bench_new.err; show $OUT/bench_new.json
MH_GEMM_ORDER=0 timeout 200 $B > $OUT/bench_order0.json 2> $OUT/bench_order0.err; show $OUT/bench_order0.json
MOTIFS_HIP_LIB=$R/$OLD MH_GEMM_ORDER=0 timeout 200 $B > $OUT/bench_r05.json 2> $OUT/bench_r05.err; show $OUT/bench_r05.json
timeout 200 $B --gemm-shapes $OUT/gemm_shapes.jsonl > $OUT/bench_new_b.json 2> $OUT/bench_new_b.err; show $OUT/bench_new_b.json
tail -3 $OUT/bench_new.err | cut -c1-200
