#!/bin/bash
# GPU call 5 (round 6): conv1_1 on the matrix cores (stem_mfma_kernel) + the new pooled-epilogue test: kernel checks, trunk /
# cfg1 / cfg2 parity, bench A/B against MH_STEM=valu, then the whole -m gpu suite on this tree
set -u
OUT=gpurun_out/r06_c5; mkdir -p $OUT; R=$PWD
LIB=neural-motifs_amd/csrc/libmotifs_hip.so
( timeout 600 tools/_bin/pl_check $LIB --conv ) > $OUT/conv_check.jsonl 2>&1; grep -c '"ok": true' $OUT/conv_check.jsonl; grep -E '"ok": false|error|summary' $OUT/conv_check.jsonl | cut -c1-300 | head -20
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "trunk or vgg or pooled or conv" > $OUT/tests_trunk.log 2>&1; grep -E "passed|failed|rror" $OUT/tests_trunk.log | tail -3 | cut -c1-300
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
MH_STEM=valu timeout 200 $B > $OUT/bench_stem_valu.json 2> /dev/null; show $OUT/bench_stem_valu.json
timeout 200 $B > $OUT/bench_new_b.json 2> /dev/null; show $OUT/bench_new_b.json
timeout 2400 python -m pytest tests -x -q -m gpu > $OUT/tests_all.log 2>&1; grep -E "passed|failed|rror" $OUT/tests_all.log | tail -5 | cut -c1-300
