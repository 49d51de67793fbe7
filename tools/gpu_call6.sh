#!/bin/bash
# round-2 GPU call 6: whole GPU suite file by file on the current tree, smoke, default bench, the changed cfg3 / cfg5 bench
# workloads, kernel stats + kernel TRACE of the bench step (for tools/trace_gaps.py)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r02_c6; mkdir -p $O
export TMPDIR=/tmp
for n in test_gpu_ops test_gpu_sgdet test_gpu_model test_gpu_configs test_gpu_baselines test_gpu_dist; do
  ( timeout 900 python -m pytest tests/$n.py -m gpu -q -s 2>&1 ) > $O/$n.log 2>&1
  echo "== $n: $(grep -E ' passed| failed|Aborted|Memory access fault' $O/$n.log | tail -2 | tr '\n' ' ')"
  grep -E "^FAILED|^E   " $O/$n.log | head -12
done
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" ) > $O/smoke.log 2>&1; tail -2 $O/smoke.log
( timeout 500 python bench.py --steps 20 --warmup 5 ) > $O/bench.log 2>&1
echo "== bench"; tail -1 $O/bench.log | cut -c1-1500
for c in cfg3 cfg5 cfg1; do
  ( timeout 300 python bench.py --config $c --no-cpu-baseline ) > $O/bench_$c.log 2>&1
  echo "== $c"; tail -1 $O/bench_$c.log | cut -c1-700
done
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $OLDPWD/bench.py --steps 8 --warmup 3 --no-cpu-baseline > $OLDPWD/$O/prof_bench.log 2>&1
cd $OLDPWD; cp $(ls /tmp/prof/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv 2>/dev/null
cp $(ls /tmp/prof/*/*kernel_trace.csv | head -1) $O/kernel_trace.csv 2>/dev/null
python tools/trace_gaps.py $O/kernel_trace.csv --steps 3 --top 40 > $O/trace_gaps.txt 2>&1; head -60 $O/trace_gaps.txt
