#!/bin/bash
# GPU call 7 (round 6): the matrix-core stem with its taps loaded one tile ahead; per-kernel durations + accuracy + bench A/B
set -u
OUT=gpurun_out/r06_c7; mkdir -p $OUT; R=$PWD
LIB=$R/neural-motifs_amd/csrc/libmotifs_hip.so
tools/_bin/pl_check $LIB --stem; MH_STEM=valu tools/_bin/pl_check $LIB --stem; tools/_bin/pl_check $LIB --stem
( timeout 600 tools/_bin/pl_check $LIB --conv ) > $OUT/conv_check.jsonl 2>&1; grep -c '"ok": true' $OUT/conv_check.jsonl; grep -E '"ok": false|error|summary' $OUT/conv_check.jsonl | cut -c1-300 | head
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "trunk or vgg" > $OUT/tests_trunk.log 2>&1; grep -E "passed|failed|rror" $OUT/tests_trunk.log | tail -3 | cut -c1-300
show() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], round(d['value'],1), d['unit'], 'ms', round(d['ms_per_step'],2), 'p50', d.get('ms_per_step_p50'),
          'gemm', round(d['roofline_gemm']['frac'],3), round(d['roofline_gemm']['ms_per_step'],2), 'imgs', round(d['roofline_gemm']['products_on_images']['frac'],3),
          'conv', round(d['roofline_conv']['frac'],3), 'trunk', round(d['roofline']['frac_trunk_only'],3), round(d['roofline_conv']['trunk_only']['ms_per_step'],2), 'cal', round(d['calibration']['plane_gemm_4096_tflops']),
          'act_planes', round(d['hbm_kernels'].get('act_planes',{}).get('ms_per_step',0),3), 'stem', round(d['hbm_kernels'].get('stem_to_image',{}).get('ms_per_step',0),3))
except Exception as e:
    print(sys.argv[1], 'unreadable', e)
PY
}
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --meter-every 2"
timeout 200 $B > $OUT/bench_new.json 2> $OUT/bench_new.err; show $OUT/bench_new.json
MH_STEM=valu timeout 200 $B > $OUT/bench_stem_valu.json 2> /dev/null; show $OUT/bench_stem_valu.json
timeout 200 $B > $OUT/bench_new_b.json 2> /dev/null; show $OUT/bench_new_b.json
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ps; timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps -- $R/tools/_bin/pl_check $LIB --stem > /dev/null 2>&1
cut -d, -f1-4,6,7 $(ls /tmp/ps/*/*kernel_stats.csv | head -1) | cut -c1-200
