#!/bin/bash
# Fabric-side traffic of the plane trunk (run on the GPU box): rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE
# passes (gfx950: TCC has 4 slots, FETCH_SIZE takes 3) over tools/_bin/pl_check --conv-replay = the 12 trunk launches of one
# bench step; tools/r04/traffic_summary.py joins them into profiles/r04_conv_traffic_summary.json.
#   tools/r03/traffic.sh <out dir under gpurun_out>
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; OUT=$R/gpurun_out/$1; mkdir -p $OUT
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/traffic_$c -- $R/tools/_bin/pl_check $R/neural-motifs_amd/csrc/libmotifs_hip.so --conv-replay > $OUT/replay_$c.jsonl 2> /tmp/traffic_$c.log )
  f=$(ls /tmp/traffic_$c/*/*counter_collection.csv 2>/dev/null | head -1)
  if [ -n "$f" ]; then cp $f $OUT/conv_$c.csv; else echo "$c failed"; tail -5 /tmp/traffic_$c.log; fi
done
python tools/r04/traffic_summary.py $OUT/conv_FETCH_SIZE.csv $OUT/conv_WRITE_SIZE.csv $OUT/replay_FETCH_SIZE.jsonl > $OUT/conv_traffic_summary.json && cat $OUT/conv_traffic_summary.json | cut -c1-600
