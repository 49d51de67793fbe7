#!/bin/bash
# round-2 GPU call 8: pinned asynchronous H2D copies (h2d) on top of the host mirrors -- parity, bench + host profile, trace
# kernel trace + idle-gap analysis
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r02_c8; mkdir -p $O
export TMPDIR=/tmp
for n in test_gpu_model test_gpu_configs test_gpu_sgdet test_gpu_baselines; do
  ( timeout 900 python -m pytest tests/$n.py -m gpu -q -s 2>&1 ) > $O/$n.log 2>&1
  echo "== $n: $(grep -E ' passed| failed|Aborted|Memory access fault' $O/$n.log | tail -2 | tr '\n' ' ')"
  grep -E "^FAILED|^E   " $O/$n.log | head -12 | cut -c1-300
done
( timeout 500 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --host-profile ) > $O/bench.log 2> $O/host_profile.txt
echo "== bench"; tail -1 $O/bench.log | cut -c1-400
head -70 $O/host_profile.txt | cut -c1-200
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $OLDPWD/bench.py --steps 8 --warmup 3 --no-cpu-baseline > $OLDPWD/$O/prof_bench.log 2>&1
cd $OLDPWD; cp $(ls /tmp/prof/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv 2>/dev/null
cp $(ls /tmp/prof/*/*kernel_trace.csv | head -1) $O/kernel_trace.csv 2>/dev/null
python tools/trace_gaps.py $O/kernel_trace.csv --steps 3 --top 25 > $O/trace_gaps.txt 2>&1; head -8 $O/trace_gaps.txt; tail -26 $O/trace_gaps.txt
( timeout 300 python bench.py --config cfg3 --no-cpu-baseline ) > $O/bench_cfg3.log 2>&1
echo "== cfg3"; tail -1 $O/bench_cfg3.log | cut -c1-900
