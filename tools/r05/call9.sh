#!/bin/bash
# GPU call 9 (round 5): the recorded cfg2 line refreshed on the final tree (1-D parameters reduced in ranges: bucket_mb), the RCCL /
# decoder tests touched since call 7, and rocprofv3 per-kernel statistics of the cfg4 (ResNet-101) step
set -u
OUT=gpurun_out/r05_c9; mkdir -p $OUT; R=$PWD
timeout 600 python -m pytest tests/test_gpu_dist.py tests/test_gpu_ops.py -x -q -m gpu -k "rccl or reducer or plain_lstm or decoder or hwlstm" > $OUT/tests.log 2>&1; grep -E "passed|failed" $OUT/tests.log | tail -2 | cut -c1-300
timeout 400 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_c9/bench.json').read().strip().splitlines()[-1])
print('cfg2', round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'unmetered', round(d['unmetered']['value'],1), 'cal', round(d['calibration']['plane_gemm_4096_tflops']), 'max unit MB', max(d['scaling_diagnostics']['bucket_mb']), d['main_stream_segments'])
PY
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_cfg4 -- python $R/bench.py --config cfg4 --steps 6 --warmup 3 --no-cpu-baseline > $R/$OUT/prof_cfg4.log 2>&1 )
cp $(ls /tmp/prof_cfg4/*/*kernel_stats.csv | head -1) $OUT/kernel_stats_cfg4.csv 2>/dev/null
tail -1 $OUT/prof_cfg4.log | cut -c1-200
