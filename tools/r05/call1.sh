#!/bin/bash
# GPU call 1 (round 5): the small-product engine (bf16x6, one launch per product, split-K reduced by the last block of a tile) --
# its unit tests, the LSTM / decoder / Linear tests that run on it, the cfg1 / cfg2 full-size parity tests, then the cfg2 bench line
# A/B against round 4's evaluation of the same products (MH_SMALL_GEMM=f16x3: row-maxima pass + f16x3 + reduce launch), the
# product shapes with their times, and the kernel statistics of the step.
set -u
OUT=gpurun_out/r05_c1; mkdir -p $OUT; R=$PWD
timeout 400 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "gemm or small_product or linear or hwlstm or decoder or packed_recurrence or lstm" > $OUT/ops_tests.log 2>&1; grep -E "passed|failed|rror" $OUT/ops_tests.log | tail -3 | cut -c1-300
timeout 400 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "cfg1 or cfg2" > $OUT/cfg_tests.log 2>&1; grep -E "passed|failed|rror" $OUT/cfg_tests.log | tail -3 | cut -c1-300
show() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], round(d['value'],1), d['unit'], 'ms', round(d['ms_per_step'],2), 'p50', d.get('ms_per_step_p50'),
          'gemm', round(d['roofline_gemm']['frac'],3), round(d['roofline_gemm']['ms_per_step'],2), 'conv', round(d['roofline']['frac'],3),
          'cal', round(d['calibration']['plane_gemm_4096_tflops']))
except Exception as e:
    print(sys.argv[1], 'unreadable', e)
PY
}
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --gemm-shapes $OUT/gemm_shapes.jsonl > $OUT/bench_bf16x6.json 2> $OUT/bench_bf16x6.err; show $OUT/bench_bf16x6.json
MH_SMALL_GEMM=f16x3 timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_f16x3.json 2> $OUT/bench_f16x3.err; show $OUT/bench_f16x3.json
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_bf16x6_b.json 2> $OUT/bench_bf16x6_b.err; show $OUT/bench_bf16x6_b.json
( cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_cfg2 -- python $R/bench.py --steps 8 --warmup 4 --no-cpu-baseline --h2d-steps 0 --meter-every 1000 > $R/$OUT/prof_cfg2.log 2>&1 )
cp $(ls /tmp/prof_cfg2/*/*kernel_stats.csv | head -1) $OUT/kernel_stats_cfg2.csv 2>/dev/null
T=$(ls /tmp/prof_cfg2/*/*kernel_trace.csv | head -1)
python tools/trace_gaps.py $T --steps 3 --top 8 > $OUT/trace_gaps_cfg2.txt 2>&1; head -8 $OUT/trace_gaps_cfg2.txt | cut -c1-200
python tools/r04/step_timeline.py $T > $OUT/step_timeline.txt 2>&1; tail -5 $OUT/step_timeline.txt | cut -c1-200
