#!/bin/bash
# GPU call 13 (round 6): tower test (small draw), the rest of the -m gpu suite from the tower test on, bench with the context-wait
# segment, per-layer PMC table + fabric traffic of the trunk's launches, kernel statistics + unfolded timeline of the step
set -u
OUT=gpurun_out/r06_c13; mkdir -p $OUT; R=$PWD
LIB=$R/neural-motifs_amd/csrc/libmotifs_hip.so
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "tower" -s > $OUT/tests_tower.log 2>&1; grep -E "passed|failed|rror|seed" $OUT/tests_tower.log | tail -14 | cut -c1-200
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
timeout 400 bash tools/r04/pmc.sh r06_c13/pmc_conv "conv3x3" $R/tools/_bin/pl_check $LIB --conv-replay 2>&1 | tail -2 | cut -c1-300
( $R/tools/_bin/pl_check $LIB --conv-replay ) > $OUT/replay.jsonl 2>&1
python tools/r06/pmc_table.py $OUT/pmc_conv $OUT/replay.jsonl > $OUT/pmc_ring_table.txt 2> $OUT/pmc_table.err; tail -14 $OUT/pmc_ring_table.txt | cut -c1-200; tail -2 $OUT/pmc_table.err
timeout 300 bash tools/r04/traffic.sh r06_c13/traffic 2>&1 | tail -1 | cut -c1-400
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_cfg2 -- python $R/bench.py --steps 8 --warmup 4 --no-cpu-baseline --h2d-steps 0 --meter-every 1000 > $R/$OUT/prof_cfg2.log 2>&1 )
cp $(ls /tmp/prof_cfg2/*/*kernel_stats.csv | head -1) $OUT/kernel_stats_cfg2.csv 2>/dev/null
T=$(ls /tmp/prof_cfg2/*/*kernel_trace.csv | head -1)
python tools/trace_gaps.py $T --steps 3 --top 8 > $OUT/trace_gaps_cfg2.txt 2>&1; head -3 $OUT/trace_gaps_cfg2.txt | cut -c1-200
python tools/r04/step_timeline.py $T > $OUT/step_timeline.txt 2>&1
python tools/r04/step_timeline.py $T --all > $OUT/step_launches.txt 2>&1
grep -E "tower" $OUT/kernel_stats_cfg2.csv | cut -d, -f1-4 | cut -c1-160
timeout 2400 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_ops.py::test_mask_tower_first_convolution_in_its_three_forms_agree > $OUT/tests_all.log 2>&1; grep -E "passed|failed|rror" $OUT/tests_all.log | tail -5 | cut -c1-300
