#!/bin/bash
# GPU call 18 (round 6): (a) union-box geometry on the host + the fused relation tail: parity tests, bench A/B against the switches off;
# (b) are cfg1 / cfg3 slower than round 5's tree on the SAME box? (_ab/r05 = the tree of 7c07bb7 with its own library)
set -u
OUT=gpurun_out/r06_c18; mkdir -p $OUT; R=$PWD
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "pair_product or tower" > $OUT/tests_ops.log 2>&1; grep -E "passed|failed|rror" $OUT/tests_ops.log | tail -4 | cut -c1-300
timeout 2400 python -m pytest tests/test_gpu_configs.py tests/test_gpu_model.py tests/test_gpu_baselines.py -x -q -m gpu > $OUT/tests_model.log 2>&1; grep -E "passed|failed|rror" $OUT/tests_model.log | tail -4 | cut -c1-300
timeout 1800 python -m pytest tests/test_gpu_sgdet.py tests/test_gpu_dist.py -x -q -m gpu > $OUT/tests_sgdet.log 2>&1; grep -E "passed|failed|rror" $OUT/tests_sgdet.log | tail -4 | cut -c1-300
show() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], round(d['value'],1), d['unit'], 'ms', round(d['ms_per_step'],2), 'p50', d.get('ms_per_step_p50'), 'cal', round(d['calibration']['plane_gemm_4096_tflops']),
          'seg', {k: round(v, 2) for k, v in d['main_stream_segments'].items() if k.endswith('_ms')}, 'host', d['step_ms']['host_p50'])
except Exception as e:
    print(sys.argv[1], 'unreadable', e)
PY
}
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline"
timeout 200 $B > $OUT/bench_new.json 2> $OUT/bench_new.err; show $OUT/bench_new.json
MOTIFS_HOST_GEOMETRY=0 MOTIFS_PAIR_PRODUCT=0 timeout 200 $B > $OUT/bench_off.json 2> /dev/null; show $OUT/bench_off.json
MOTIFS_PAIR_PRODUCT=0 timeout 200 $B > $OUT/bench_geom_only.json 2> /dev/null; show $OUT/bench_geom_only.json
timeout 200 $B > $OUT/bench_new_b.json 2> /dev/null; show $OUT/bench_new_b.json
MOTIFS_HOST_GEOMETRY=0 MOTIFS_PAIR_PRODUCT=0 timeout 200 $B > $OUT/bench_off_b.json 2> /dev/null; show $OUT/bench_off_b.json
one() { ( cd $1 && timeout 400 python bench.py --config $2 --steps 10 --warmup 3 2>/dev/null | tail -1 ) > $OUT/$3.json; python -c "
import json; d=json.loads(open('$OUT/$3.json').read()); print('$3', round(d['value'],1), d['unit'], round(d['ms_per_step'],2), 'ms')"; }
for c in cfg1 cfg3; do one $R $c new_$c; one $R/_ab/r05 $c r05_$c; one $R $c new_${c}_b; one $R/_ab/r05 $c r05_${c}_b; done
