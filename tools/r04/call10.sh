#!/bin/bash
# GPU call 10 (round 4): the step's timeline with the union-box backward issued first (MOTIFS_LATE_VR=auto); sgdet kink location
set -u
OUT=gpurun_out/r04_c10; mkdir -p $OUT; R=$PWD
( cd /tmp && export TMPDIR=/tmp && MOTIFS_LATE_VR=auto timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $R/bench.py --steps 8 --warmup 4 --no-cpu-baseline --h2d-steps 0 --meter-every 1000 > $R/$OUT/prof_bench.log 2>&1 )
T=$(ls /tmp/prof/*/*kernel_trace.csv | head -1)
python tools/trace_gaps.py $T --steps 3 --top 4 > $OUT/trace_gaps.txt 2>&1; head -7 $OUT/trace_gaps.txt | cut -c1-220
python tools/r04/step_timeline.py $T --step -2 > $OUT/step_timeline.txt 2>&1; head -4 $OUT/step_timeline.txt
grep -E "hw_layer|gemm_ring|pl::gemm_kernel|multi_sgd|conv3x3_nhwc|wgrad" $OUT/step_timeline.txt | grep -v "^  q" | cut -c1-110
for late in 0 auto; do MOTIFS_LATE_VR=$late timeout 100 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --h2d-steps 0 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('late=$late', round(d['value'],1), 'img/s p50', d['ms_per_step_p50'], 'lstm fwd/bwd us', round(d['hbm_kernels']['lstm_fwd']['us_per_call']), round(d['hbm_kernels']['lstm_bwd']['us_per_call']))"; done
timeout 300 python -m pytest tests/test_gpu_sgdet.py -x -q -s -k "test_sgdet_train_step_parity" > $OUT/sgdet.log 2>&1
grep -E "passed|failed" $OUT/sgdet.log | tail -1; grep -E "kink site context.pos" $OUT/sgdet.log | cut -c1-250
