#!/bin/bash
# GPU call 26: plane conv with the per-k-tile offset table in LDS -- accuracy (pl_check --conv), trunk test, smoke, bench
set -u
OUT=gpurun_out/r03_c26; mkdir -p $OUT
( timeout 90 tools/_bin/pl_check neural-motifs_amd/csrc/libmotifs_hip.so --conv ) > $OUT/pl_conv_check.jsonl 2>&1; tail -1 $OUT/pl_conv_check.jsonl; grep -c '"ok": false' $OUT/pl_conv_check.jsonl
timeout 100 python -m pytest tests/test_gpu_ops.py -x -q -k "vgg_trunk_on_the_plane or next_to_the_in_loop" 2>&1 | tail -2
timeout 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench.json
python -c "
import json; d=json.loads(open('$OUT/bench.json').read()); t=d['roofline']['trunk_only']; print('bench', round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms; calib', round(d['calibration']['plane_gemm_4096_tflops'],1), '; conv', round(d['roofline']['achieved'],1), 'trunk', round(t['tflops'],1), 'TF', round(t['ms_per_step'],2), 'ms; gemm', round(d['roofline_gemm']['ms_per_step'],2), 'ms')"
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
