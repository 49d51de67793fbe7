#!/bin/bash
# round-2 GPU call 4: every GPU test file in its own process (a device fault in one must not hide the others), bench, profile
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r02_c4; mkdir -p $O
export TMPDIR=/tmp
for f in tests/test_gpu_*.py; do
  n=$(basename $f .py)
  ( timeout 900 python -m pytest $f -m gpu -q -s 2>&1 ) > $O/$n.log 2>&1
  echo "== $n: $(grep -E ' passed| failed|Aborted|Memory access fault' $O/$n.log | tail -2 | tr '\n' ' ')"
  grep -E "^FAILED|^E   " $O/$n.log | head -12
done
timeout 200 python tools/gpu_perf_conv.py all > $O/perf_default.log 2>&1
echo "== perf_default"; grep -v "^{" $O/perf_default.log | grep -E "TRUNK fp32 |GEMM|conv" | cut -c1-60
( timeout 500 python bench.py --steps 20 --warmup 5 ) > $O/bench.log 2>&1
echo "== bench"; tail -1 $O/bench.log | cut -c1-3500
( timeout 300 python bench.py --config cfg5 --steps 10 --warmup 3 ) > $O/bench_cfg5.log 2>&1; echo "== bench cfg5"; tail -2 $O/bench_cfg5.log | cut -c1-1200
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $OLDPWD/bench.py --steps 8 --warmup 3 --no-cpu-baseline > $OLDPWD/$O/prof_bench.log 2>&1
cd $OLDPWD; cp $(ls /tmp/prof/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv 2>/dev/null; head -14 $O/kernel_stats.csv | cut -c1-150
