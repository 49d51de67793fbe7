#!/bin/bash
# round-2 GPU call 9: pinned ring for the small uploads -- bench A/B (ring / torch pinned allocator / pageable), host profile, trace
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r02_c9; mkdir -p $O
export TMPDIR=/tmp
for how in ring alloc pageable; do
  ( MOTIFS_H2D=$how timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --host-profile ) > $O/bench_$how.log 2> $O/host_profile_$how.txt
  echo "== bench H2D=$how: $(tail -1 $O/bench_$how.log | cut -c1-160)"; grep "host enqueue" $O/host_profile_$how.txt
done
( MOTIFS_OVERLAP=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --host-profile ) > $O/bench_ring_1stream.log 2> $O/host_profile_ring_1stream.txt
echo "== bench ring, one stream: $(tail -1 $O/bench_ring_1stream.log | cut -c1-160)"; grep "host enqueue" $O/host_profile_ring_1stream.txt
( PYTORCH_NO_HIP_MEMORY_CACHING=0 PYTORCH_HIP_ALLOC_CONF=expandable_segments:True timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --host-profile ) > $O/bench_ring_expand.log 2> $O/host_profile_ring_expand.txt
echo "== bench ring, expandable segments: $(tail -1 $O/bench_ring_expand.log | cut -c1-160)"; grep "host enqueue" $O/host_profile_ring_expand.txt
head -30 $O/host_profile_ring.txt | cut -c1-180
for n in test_gpu_model test_gpu_ops; do
  ( timeout 900 python -m pytest tests/$n.py -m gpu -q -s 2>&1 ) > $O/$n.log 2>&1
  echo "== $n: $(grep -E ' passed| failed|Aborted|Memory access fault' $O/$n.log | tail -2 | tr '\n' ' ')"
  grep -E "^FAILED|^E   " $O/$n.log | head -12 | cut -c1-300
done
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $OLDPWD/bench.py --steps 8 --warmup 3 --no-cpu-baseline > $OLDPWD/$O/prof_bench.log 2>&1
cd $OLDPWD; cp $(ls /tmp/prof/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv 2>/dev/null
cp $(ls /tmp/prof/*/*kernel_trace.csv | head -1) $O/kernel_trace.csv 2>/dev/null
python tools/trace_gaps.py $O/kernel_trace.csv --steps 3 --top 25 > $O/trace_gaps.txt 2>&1; head -8 $O/trace_gaps.txt; tail -26 $O/trace_gaps.txt
( timeout 300 python bench.py --config cfg3 --no-cpu-baseline ) > $O/bench_cfg3.log 2>&1
echo "== cfg3"; tail -1 $O/bench_cfg3.log | cut -c1-900
