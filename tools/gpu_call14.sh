#!/bin/bash
# round-2 GPU call 14 (final state of the round): whole GPU suite file by file, smoke, default bench (with the CPU baseline),
# kernel stats + trace of the bench step, secondary configs, PMC counters of the two MFMA kernels on the f16x3 build
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r02_c14; mkdir -p $O
export TMPDIR=/tmp
for n in test_gpu_ops test_gpu_sgdet test_gpu_model test_gpu_configs test_gpu_baselines test_gpu_dist; do
  ( timeout 900 python -m pytest tests/$n.py -m gpu -q -s 2>&1 ) > $O/$n.log 2>&1
  echo "== $n: $(grep -E ' passed| failed|Aborted|Memory access fault' $O/$n.log | tail -2 | tr '\n' ' ')"
  grep -E "^FAILED|^E   " $O/$n.log | head -12 | cut -c1-300
done
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" ) > $O/smoke.log 2>&1; tail -2 $O/smoke.log
( timeout 500 python bench.py --steps 30 --warmup 6 ) > $O/bench.log 2>&1
echo "== bench"; tail -1 $O/bench.log | cut -c1-2200
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $OLDPWD/bench.py --steps 8 --warmup 4 --no-cpu-baseline > $OLDPWD/$O/prof_bench.log 2>&1
cd $OLDPWD; cp $(ls /tmp/prof/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv 2>/dev/null
cp $(ls /tmp/prof/*/*kernel_trace.csv | head -1) $O/kernel_trace.csv 2>/dev/null
python tools/trace_gaps.py $O/kernel_trace.csv --steps 3 --top 25 > $O/trace_gaps.txt 2>&1; head -8 $O/trace_gaps.txt
for c in cfg3 cfg1; do
  ( timeout 300 python bench.py --config $c --no-cpu-baseline ) > $O/bench_$c.log 2>&1
  echo "== $c"; tail -1 $O/bench_$c.log | cut -c1-400
done
timeout 400 bash tools/pmc_run.sh > $O/pmc.log 2>&1; cp gpurun_out/pmc/set*.csv $O/ 2>/dev/null; tail -12 $O/pmc.log | cut -c1-1500
( timeout 200 python bench.py --config cfg5 --no-cpu-baseline ) > $O/bench_cfg5.log 2>&1
echo "== cfg5"; tail -1 $O/bench_cfg5.log | cut -c1-400
