#!/bin/bash
# GPU call 4 (round 6): call 3 again with the scale words of a sliced image written by the tile-0 finisher, the image epilogues evaluated at the output scale, 12 warm-up launches per timing
# the ticket) + the first-round stagger experiment (pl_check --conv-r06: debug flags 0x100 x n)
set -u
OUT=gpurun_out/r06_c4; mkdir -p $OUT; R=$PWD
LIB=neural-motifs_amd/csrc/libmotifs_hip.so
( timeout 600 tools/_bin/pl_check $LIB --conv ) > $OUT/conv_check.jsonl 2>&1; grep -c '"ok": true' $OUT/conv_check.jsonl; grep -E '"ok": false|error|summary' $OUT/conv_check.jsonl | cut -c1-300 | head -20
( timeout 600 tools/_bin/pl_check $LIB --conv ) > $OUT/conv_check_b.jsonl 2>&1; grep -E '"ok": false|error|summary' $OUT/conv_check_b.jsonl | cut -c1-300 | head -20
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_ops.py -x -q -m gpu -k "trunk or vgg or plane or conv or small_product or decoder or lstm" > $OUT/tests_trunk.log 2>&1; grep -E "passed|failed|rror" $OUT/tests_trunk.log | tail -3 | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "cfg1 or cfg2" > $OUT/tests_cfg.log 2>&1; grep -E "passed|failed|rror" $OUT/tests_cfg.log | tail -3 | cut -c1-300
( timeout 900 tools/_bin/pl_check $LIB --conv-r06 ) > $OUT/conv_r06.jsonl 2>&1
python - <<'PY'
import json
for l in open('gpurun_out/r06_c4/conv_r06.jsonl'):
    if not l.startswith('{'): print(l.strip()[:200]); continue
    d = json.loads(l)
    if 'debug_flags' in d: print('== debug_flags', hex(d['debug_flags'])); continue
    print(d['case'], 'shape', d['shape'], 'sk', d['splitk'], 'fp32', d['ms'], d['tflops'], 'img', d['ms_image_out'], d['tflops_image_out'], 'pool', d['ms_pooled_image_out'], d['tflops_pooled_image_out'])
PY
show() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], round(d['value'],1), d['unit'], 'ms', round(d['ms_per_step'],2), 'p50', d.get('ms_per_step_p50'),
          'gemm', round(d['roofline_gemm']['frac'],3), round(d['roofline_gemm']['ms_per_step'],2), 'imgs', round(d['roofline_gemm']['products_on_images']['frac'],3),
          'conv', round(d['roofline_conv']['frac'],3), 'trunk', round(d['roofline']['frac_trunk_only'],3), 'cal', round(d['calibration']['plane_gemm_4096_tflops']),
          'act_planes', round(d['hbm_kernels'].get('act_planes',{}).get('ms_per_step',0),3), 'stem', round(d['hbm_kernels'].get('stem_to_image',{}).get('ms_per_step',0),3))
except Exception as e:
    print(sys.argv[1], 'unreadable', e)
PY
}
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline"
timeout 200 $B > $OUT/bench_new.json 2> $OUT/bench_new.err; show $OUT/bench_new.json
MOTIFS_TRUNK_POOL=converter timeout 200 $B > $OUT/bench_pool_converter.json 2> $OUT/bench_pool_converter.err; show $OUT/bench_pool_converter.json; tail -5 $OUT/bench_pool_converter.err | cut -c1-300
MH_PLCONV_ONE_ROUND=1 timeout 200 $B > $OUT/bench_conv5_whole.json 2> /dev/null; show $OUT/bench_conv5_whole.json
timeout 200 $B > $OUT/bench_new_b.json 2> /dev/null; show $OUT/bench_new_b.json
