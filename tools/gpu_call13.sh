#!/bin/bash
# round-2 GPU call 13: run-ahead bound A/B on the stall-free asynchronous step
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r02_c13; mkdir -p $O
export TMPDIR=/tmp
for a in 3 2 -1 1; do
  ( MOTIFS_MAX_AHEAD=$a timeout 200 python bench.py --steps 40 --warmup 6 --no-cpu-baseline ) > $O/bench_ahead$a.log 2>&1
  echo "== bench max_ahead=$a: $(tail -1 $O/bench_ahead$a.log | cut -c1-150)"
done
( MOTIFS_MAX_AHEAD=3 MOTIFS_OVERLAP=0 timeout 200 python bench.py --steps 40 --warmup 6 --no-cpu-baseline ) > $O/bench_ahead3_1stream.log 2>&1
echo "== bench max_ahead=3, one stream: $(tail -1 $O/bench_ahead3_1stream.log | cut -c1-150)"
