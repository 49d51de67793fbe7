#!/bin/bash
# GPU call 6 (round 6): where the stem's 0.4 ms goes -- per-kernel durations of mh_stem_to_image (memset, image maxima, conv) for
# the matrix-core and the VALU kernel
set -u
OUT=gpurun_out/r06_c6; mkdir -p $OUT; R=$PWD
LIB=$R/neural-motifs_amd/csrc/libmotifs_hip.so
tools/_bin/pl_check $LIB --stem; MH_STEM=valu tools/_bin/pl_check $LIB --stem
cd /tmp && export TMPDIR=/tmp
for v in mfma valu; do
  rm -rf /tmp/ps; MH_STEM=$v timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps -- $R/tools/_bin/pl_check $LIB --stem > /dev/null 2>&1
  cp $(ls /tmp/ps/*/*kernel_stats.csv | head -1) $R/$OUT/stem_$v.kernel_stats.csv; echo "== $v"; cut -d, -f1-4 $R/$OUT/stem_$v.kernel_stats.csv | cut -c1-160
done
