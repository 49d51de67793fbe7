#!/bin/bash
# GPU call 18 (round 4): the mask kernel with scalar-cache column boxes (bit-exact NMS tests, timing), then rocprofv3 kernel statistics of the
# cfg3 and cfg2 steps on the final tree (the committed r04 statistics predate the LSTM granules and the NMS sweep)
set -u
OUT=gpurun_out/r04_c18; mkdir -p $OUT; R=$PWD
timeout 150 python -m pytest tests/ -x -q -m gpu -k "nms or filter_det or proposal" > $OUT/nms_tests.log 2>&1; grep -E "passed|failed|rror" $OUT/nms_tests.log | tail -2 | cut -c1-200
timeout 100 python tools/r04/nms_time.py > $OUT/nms_time.jsonl 2> $OUT/nms_time.err; cut -c1-330 $OUT/nms_time.jsonl
for c in cfg3 cfg2; do
  ( cd /tmp && export TMPDIR=/tmp && timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$c -- python $R/bench.py --config $c --steps 8 --warmup 4 --no-cpu-baseline --h2d-steps 0 --meter-every 1000 > $R/$OUT/prof_$c.log 2>&1 )
  cp $(ls /tmp/prof_$c/*/*kernel_stats.csv | head -1) $OUT/kernel_stats_$c.csv 2>/dev/null
  T=$(ls /tmp/prof_$c/*/*kernel_trace.csv | head -1)
  python tools/trace_gaps.py $T --steps 3 --top 8 > $OUT/trace_gaps_$c.txt 2>&1; head -6 $OUT/trace_gaps_$c.txt | cut -c1-200
  tail -1 $OUT/prof_$c.log | cut -c1-120
done
