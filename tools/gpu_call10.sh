#!/bin/bash
# round-2 GPU call 10: bounded host run-ahead (FusedClipSGD.max_ahead) on top of the asynchronous step -- bench A/B, trace
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r02_c10; mkdir -p $O
export TMPDIR=/tmp
for a in 1 2 0; do
  ( MOTIFS_MAX_AHEAD=$a timeout 300 python bench.py --steps 30 --warmup 6 --no-cpu-baseline --host-profile ) > $O/bench_ahead$a.log 2> $O/host_profile_ahead$a.txt
  echo "== bench max_ahead=$a: $(tail -1 $O/bench_ahead$a.log | cut -c1-160)"; grep "host enqueue" $O/host_profile_ahead$a.txt
done
( MOTIFS_MAX_AHEAD=1 MOTIFS_OVERLAP=0 timeout 300 python bench.py --steps 30 --warmup 6 --no-cpu-baseline ) > $O/bench_ahead1_1stream.log 2>&1
echo "== bench max_ahead=1, one stream: $(tail -1 $O/bench_ahead1_1stream.log | cut -c1-160)"
head -28 $O/host_profile_ahead1.txt | cut -c1-180
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $OLDPWD/bench.py --steps 8 --warmup 4 --no-cpu-baseline > $OLDPWD/$O/prof_bench.log 2>&1
cd $OLDPWD; cp $(ls /tmp/prof/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv 2>/dev/null
cp $(ls /tmp/prof/*/*kernel_trace.csv | head -1) $O/kernel_trace.csv 2>/dev/null
python tools/trace_gaps.py $O/kernel_trace.csv --steps 3 --top 25 > $O/trace_gaps.txt 2>&1; head -8 $O/trace_gaps.txt; tail -26 $O/trace_gaps.txt
( timeout 300 python bench.py --config cfg3 --no-cpu-baseline ) > $O/bench_cfg3.log 2>&1
echo "== cfg3"; tail -1 $O/bench_cfg3.log | cut -c1-300
