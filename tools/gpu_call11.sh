#!/bin/bash
# round-2 GPU call 11: what stalls run_backward for 50-80 ms every few steps once the step is asynchronous? (allocator / GC / stack probes)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r02_c11; mkdir -p $O
export TMPDIR=/tmp
( timeout 300 python bench.py --steps 12 --warmup 6 --no-cpu-baseline --host-profile ) > $O/bench.log 2> $O/probe.txt
tail -1 $O/bench.log | cut -c1-160
grep -E "^probe|host enqueue" $O/probe.txt | cut -c1-1200
grep -A14 "most recent call first" $O/probe.txt | head -150 | cut -c1-150
