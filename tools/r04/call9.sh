#!/bin/bash
# GPU call 9 (round 4): sgdet parity (float64 floor for ill-conditioned gradients), cfg4 at its stated size, kernel trace of the bench step
set -u
OUT=gpurun_out/r04_c9; mkdir -p $OUT; R=$PWD
timeout 300 python -m pytest tests/test_gpu_sgdet.py -x -q -s -k "test_sgdet_train_step_parity" > $OUT/sgdet.log 2>&1
grep -E "passed|failed" $OUT/sgdet.log | tail -1; grep -E "FLOAT64|beyond" $OUT/sgdet.log | cut -c1-220
timeout 900 python -m pytest tests/test_gpu_configs.py -x -q -s -k "test_cfg4_resnet_sgcls_train_step_b6_1536_rows" > $OUT/cfg4_full.log 2>&1
grep -E "passed|failed|Error" $OUT/cfg4_full.log | tail -3 | cut -c1-300; grep -E "^cfg4" $OUT/cfg4_full.log | awk '{print $NF, $0}' | sort -g | tail -6 | cut -c1-200
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $R/bench.py --steps 8 --warmup 4 --no-cpu-baseline --h2d-steps 0 --meter-every 1000 > $R/$OUT/prof_bench.log 2>&1 )
cp $(ls /tmp/prof/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv 2>/dev/null
T=$(ls /tmp/prof/*/*kernel_trace.csv | head -1)
python tools/trace_gaps.py $T --steps 3 --top 10 > $OUT/trace_gaps.txt 2>&1; head -8 $OUT/trace_gaps.txt | cut -c1-220
python tools/r04/step_timeline.py $T --step -2 > $OUT/step_timeline.txt 2>&1; head -5 $OUT/step_timeline.txt
tail -1 $OUT/prof_bench.log | cut -c1-160
