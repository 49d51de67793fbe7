#!/bin/bash
# HBM-side traffic of the conv3x3 implicit-GEMM kernel (default) or of the fc6/fc7 GEMMs (`gemm`), one launch per bench
# shape (tools/conv_traffic.cpp / tools/gemm_traffic.cpp), collected as MI355X_MICROARCH.md prescribes: FETCH_SIZE and
# WRITE_SIZE in SEPARATE --pmc passes, counters only (no trace domain besides --kernel-trace).  Run on the GPU box from the
# repo root; results land in gpurun_out/traffic/<tool>.<tag>.<counter>.csv + <tool>.launches.jsonl
#   tools/traffic_run.sh [conv|gemm] [tag=ENV_ASSIGNMENT ...]        e.g.  tools/traffic_run.sh gemm rows=MH_GEMM_PATCH=rows
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/traffic; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
TOOL=${1:-conv}; shift
LIB=$ROOT/neural-motifs_amd/csrc/libmotifs_hip.so
VARIANTS=${@:-default=MH_NOP=1}
for v in $VARIANTS; do
  tag=${v%%=*}; assign=${v#*=}
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/tr
    env $assign timeout 60 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/tr -- $ROOT/tools/_bin/${TOOL}_traffic $LIB > $OUT/$TOOL.launches.jsonl 2> $OUT/$TOOL.$tag.$ctr.log
    cp $(ls /tmp/tr/*/*counter_collection.csv 2>/dev/null | head -1) $OUT/$TOOL.$tag.$ctr.csv 2>/dev/null || tail -3 $OUT/$TOOL.$tag.$ctr.log
  done
done
ls $OUT
