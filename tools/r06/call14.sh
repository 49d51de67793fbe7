#!/bin/bash
# GPU call 14 (round 6): warm per-layer PMC table of the trunk (three launches per layer, the last tabulated); the tower weight
# gradient with two rows in flight, 16-byte image maxima, unrolled BatchNorm finalize: tests + bench + kernel statistics
set -u
OUT=gpurun_out/r06_c14; mkdir -p $OUT; R=$PWD
LIB=$R/neural-motifs_amd/csrc/libmotifs_hip.so
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "tower or maps or bn or batch" > $OUT/tests_ops.log 2>&1; grep -E "passed|failed|rror" $OUT/tests_ops.log | tail -3 | cut -c1-200
timeout 1200 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "cfg2 or cfg1" > $OUT/tests_cfg.log 2>&1; grep -E "passed|failed|rror" $OUT/tests_cfg.log | tail -3 | cut -c1-300
show() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], round(d['value'],1), d['unit'], 'ms', round(d['ms_per_step'],2), 'p50', d.get('ms_per_step_p50'),
          'gemm', round(d['roofline_gemm']['frac'],3), round(d['roofline_gemm']['ms_per_step'],2), 'imgs', round(d['roofline_gemm']['products_on_images']['frac'],3),
          'conv', round(d['roofline_conv']['frac'],3), 'trunk', round(d['roofline']['frac_trunk_only'],3), round(d['roofline_conv']['trunk_only']['ms_per_step'],2), 'cal', round(d['calibration']['plane_gemm_4096_tflops']),
          'seg', {k: round(v, 2) for k, v in d['main_stream_segments'].items() if k.endswith('_ms')})
except Exception as e:
    print(sys.argv[1], 'unreadable', e)
PY
}
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline"
timeout 200 $B > $OUT/bench_new.json 2> $OUT/bench_new.err; show $OUT/bench_new.json
timeout 200 $B > $OUT/bench_new_b.json 2> /dev/null; show $OUT/bench_new_b.json
timeout 600 bash tools/r04/pmc.sh r06_c14/pmc_conv "conv3x3" $R/tools/_bin/pl_check $LIB --conv-replay 3 2>&1 | tail -2 | cut -c1-300
( $R/tools/_bin/pl_check $LIB --conv-replay 3 ) > $OUT/replay.jsonl 2>&1
python tools/r06/pmc_table.py $OUT/pmc_conv $OUT/replay.jsonl > $OUT/pmc_ring_table.txt 2> $OUT/pmc_table.err; tail -13 $OUT/pmc_ring_table.txt | cut -c1-200; tail -2 $OUT/pmc_table.err
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_cfg2 -- python $R/bench.py --steps 8 --warmup 4 --no-cpu-baseline --h2d-steps 0 --meter-every 1000 > $R/$OUT/prof_cfg2.log 2>&1 )
cp $(ls /tmp/prof_cfg2/*/*kernel_stats.csv | head -1) $OUT/kernel_stats_cfg2.csv 2>/dev/null
grep -E "tower|image_absmax|bn_finalize" $OUT/kernel_stats_cfg2.csv | cut -d, -f1-4 | cut -c1-160
for c in cfg4; do timeout 400 python bench.py --config $c --steps 10 --warmup 3 2>/dev/null | tail -1 > $OUT/bench_$c.json; python -c "
import json; d=json.loads(open('$OUT/bench_$c.json').read()); print('$c', round(d['value'],1), d['unit'], round(d['ms_per_step'],2), 'ms')"; done
