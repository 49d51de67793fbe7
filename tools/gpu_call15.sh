#!/bin/bash
# round-2 GPU call 15: conflict-free packed-plane staging in the conv kernel (row permutation) -- numerics + model parity + bench
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r02_c15; mkdir -p $O
export TMPDIR=/tmp
for n in test_gpu_ops test_gpu_model; do
  ( timeout 600 python -m pytest tests/$n.py -m gpu -q -s 2>&1 ) > $O/$n.log 2>&1
  echo "== $n: $(grep -E ' passed| failed|Aborted|Memory access fault' $O/$n.log | tail -2 | tr '\n' ' ')"
  grep -E "^FAILED|^E   " $O/$n.log | head -12 | cut -c1-300
done
( timeout 200 python bench.py --steps 30 --warmup 6 --no-cpu-baseline ) > $O/bench.log 2>&1
echo "== bench: $(tail -1 $O/bench.log | cut -c1-1100)"
timeout 100 python tools/gpu_perf_conv.py conv > $O/perf_conv.log 2>&1; grep -E "TRUNK|conv" $O/perf_conv.log | grep -v "^{" | cut -c1-60
