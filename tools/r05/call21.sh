#!/bin/bash
# GPU call 21 (round 5): rocprofv3 per-kernel statistics / timeline of the cfg2 step on the FINAL tree (the command of the recorded line,
# fewer steps, meters off)
set -u
OUT=gpurun_out/r05_c21; mkdir -p $OUT; R=$PWD
( cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_cfg2 -- python $R/bench.py --steps 8 --warmup 4 --no-cpu-baseline --h2d-steps 0 --meter-every 1000 > $R/$OUT/prof_cfg2.log 2>&1 )
cp $(ls /tmp/prof_cfg2/*/*kernel_stats.csv | head -1) $OUT/kernel_stats_cfg2.csv 2>/dev/null
T=$(ls /tmp/prof_cfg2/*/*kernel_trace.csv | head -1)
python tools/trace_gaps.py $T --steps 3 --top 8 > $OUT/trace_gaps_cfg2.txt 2>&1; head -4 $OUT/trace_gaps_cfg2.txt | cut -c1-200
python tools/r04/step_timeline.py $T > $OUT/step_timeline.txt 2>&1; head -4 $OUT/step_timeline.txt | cut -c1-200
head -8 $OUT/kernel_stats_cfg2.csv | cut -c1-160
