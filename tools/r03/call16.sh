#!/bin/bash
set -u
OUT=gpurun_out/r03_c16; mkdir -p $OUT
export DIAG7_VICTIMS="roi_align_fwd nhwc"
MOTIFS_HIP_LIB=$PWD/neural-motifs_amd/csrc/_variants/roiB/libmotifs_hip.so timeout 200 python tools/r03/diag7_pairs.py 2>&1 | grep -E "^PAIR|Error|error" >> $OUT/diag7.log
timeout 200 python tools/r03/diag7_pairs.py 2>&1 | grep -E "^PAIR|Error|error" >> $OUT/diag7.log
cat $OUT/diag7.log | cut -c1-220
timeout 240 python tools/r03/diag3_streams.py roi_single_address_stores 2>&1 | grep VARIANT | tee $OUT/diag3.log
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "roi or RoI" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_configs.py -x -q 2>&1 | tail -4 | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_sgdet.py -x -q 2>&1 | tail -4 | cut -c1-300
