#!/bin/bash
# GPU call 9 (round 6): the stem with two blocks per CU in the step (bench A/B against MH_STEM_BLOCKS=8), then rocprofv3 per-kernel
# statistics / timeline of the cfg2 step on this tree (the baseline of the round's remaining work)
set -u
OUT=gpurun_out/r06_c9; mkdir -p $OUT; R=$PWD
LIB=$R/neural-motifs_amd/csrc/libmotifs_hip.so
tools/_bin/pl_check $LIB --stem; MH_STEM_BLOCKS=8 tools/_bin/pl_check $LIB --stem; tools/_bin/pl_check $LIB --stem
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "trunk or vgg or stem" > $OUT/tests_trunk.log 2>&1; grep -E "passed|failed|rror" $OUT/tests_trunk.log | tail -3 | cut -c1-300
show() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], round(d['value'],1), d['unit'], 'ms', round(d['ms_per_step'],2), 'p50', d.get('ms_per_step_p50'),
          'gemm', round(d['roofline_gemm']['frac'],3), round(d['roofline_gemm']['ms_per_step'],2), 'imgs', round(d['roofline_gemm']['products_on_images']['frac'],3),
          'conv', round(d['roofline_conv']['frac'],3), 'trunk', round(d['roofline']['frac_trunk_only'],3), round(d['roofline_conv']['trunk_only']['ms_per_step'],2), 'cal', round(d['calibration']['plane_gemm_4096_tflops']),
          'act_planes', round(d['hbm_kernels'].get('act_planes',{}).get('ms_per_step',0),3), 'stem', round(d['hbm_kernels'].get('stem_to_image',{}).get('ms_per_step',0),3),
          'seg', {k: round(v, 2) for k, v in d['main_stream_segments'].items() if k.endswith('_ms')})
except Exception as e:
    print(sys.argv[1], 'unreadable', e)
PY
}
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --meter-every 2"
timeout 200 $B > $OUT/bench_new.json 2> $OUT/bench_new.err; show $OUT/bench_new.json
MH_STEM_BLOCKS=8 timeout 200 $B > $OUT/bench_stem8.json 2> /dev/null; show $OUT/bench_stem8.json
MH_STEM=valu timeout 200 $B > $OUT/bench_stem_valu.json 2> /dev/null; show $OUT/bench_stem_valu.json
timeout 200 $B > $OUT/bench_new_b.json 2> /dev/null; show $OUT/bench_new_b.json
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_cfg2 -- python $R/bench.py --steps 8 --warmup 4 --no-cpu-baseline --h2d-steps 0 --meter-every 1000 > $R/$OUT/prof_cfg2.log 2>&1 )
cp $(ls /tmp/prof_cfg2/*/*kernel_stats.csv | head -1) $OUT/kernel_stats_cfg2.csv 2>/dev/null
T=$(ls /tmp/prof_cfg2/*/*kernel_trace.csv | head -1)
python tools/trace_gaps.py $T --steps 3 --top 8 > $OUT/trace_gaps_cfg2.txt 2>&1; head -4 $OUT/trace_gaps_cfg2.txt | cut -c1-200
python tools/r04/step_timeline.py $T > $OUT/step_timeline.txt 2>&1; head -4 $OUT/step_timeline.txt | cut -c1-200
head -12 $OUT/kernel_stats_cfg2.csv | cut -c1-160
