#!/bin/bash
# HBM-side traffic of the conv3x3 implicit-GEMM kernel, one launch per bench shape (tools/conv_traffic.cpp), collected
# as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes, counters only (no trace domain
# besides --kernel-trace).  Run on the GPU box from the repo root; results land in gpurun_out/traffic/.
#   tools/traffic_run.sh [lib.so ...]      default: the in-tree library
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/traffic; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
LIBS=${@:-neural-motifs_amd/csrc/libmotifs_hip.so}
for lib in $LIBS; do
  tag=$(basename $(dirname $lib))
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/tr
    timeout 8 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/tr -- $ROOT/tools/_bin/conv_traffic $ROOT/$lib > $OUT/launches.jsonl 2> $OUT/$tag.$ctr.log
    cp $(ls /tmp/tr/*/*counter_collection.csv 2>/dev/null | head -1) $OUT/$tag.$ctr.csv 2>/dev/null || tail -3 $OUT/$tag.$ctr.log
  done
done
ls $OUT
