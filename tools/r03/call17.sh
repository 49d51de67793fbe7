#!/bin/bash
set -u
OUT=gpurun_out/r03_c17; mkdir -p $OUT
export DIAG7_VICTIMS="roi_align_fwd nhwc"
V=$PWD/neural-motifs_amd/csrc/_variants
MOTIFS_HIP_LIB=$V/roiC/libmotifs_hip.so timeout 200 python tools/r03/diag7_pairs.py 2>&1 | grep -E "^PAIR|Error|error" >> $OUT/diag7.log
MH_ROI_NOVEC=1 timeout 200 python tools/r03/diag7_pairs.py 2>&1 | grep -E "^PAIR|Error|error" | sed 's/lib=default/lib=default+novec/' >> $OUT/diag7.log
MH_ROI_NOVEC=1 MOTIFS_HIP_LIB=$V/roiC/libmotifs_hip.so timeout 200 python tools/r03/diag7_pairs.py 2>&1 | grep -E "^PAIR|Error|error" | sed 's/lib=roiC/lib=roiC+novec/' >> $OUT/diag7.log
cat $OUT/diag7.log | cut -c1-220
