#!/bin/bash
# round-3 GPU call 5: plane trunk after the contiguous-range fix of the maxima reductions: conv5 speed, model parity, bench,
# kernel stats
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=gpurun_out/r03_c5; mkdir -p $O
export TMPDIR=/tmp
( timeout 600 tools/_bin/pl_check neural-motifs_amd/csrc/libmotifs_hip.so --conv --speed ) > $O/pl_conv_check.jsonl 2>&1
echo "== pl_check conv rc=$?"; grep -c '"ok": true' $O/pl_conv_check.jsonl; grep '"ok": false\|error\|summary' $O/pl_conv_check.jsonl | head -20 | cut -c1-400
grep "conv speed" $O/pl_conv_check.jsonl | grep "conv5\|conv1_2" | cut -c1-200
for n in test_gpu_ops test_gpu_model test_gpu_configs; do
  ( timeout 900 python -m pytest tests/$n.py -m gpu -q -s 2>&1 ) > $O/$n.log 2>&1
  echo "== $n: $(grep -E ' passed| failed|Aborted|Memory access fault' $O/$n.log | tail -2 | tr '\n' ' ')"
  grep -E "^FAILED|^E   " $O/$n.log | head -12 | cut -c1-300
done
( timeout 300 python bench.py --steps 30 --warmup 6 --no-cpu-baseline ) > $O/bench.log 2>&1
echo "== bench (plane trunk): $(tail -1 $O/bench.log | cut -c1-400)"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $R/bench.py --steps 8 --warmup 4 --no-cpu-baseline > $R/$O/prof_bench.log 2>&1 )
cp $(ls /tmp/prof/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv 2>/dev/null
cp $(ls /tmp/prof/*/*kernel_trace.csv | head -1) $O/kernel_trace.csv 2>/dev/null
head -40 $O/kernel_stats.csv | cut -c1-150
python tools/trace_gaps.py $O/kernel_trace.csv --steps 3 --top 12 > $O/trace_gaps.txt 2>&1; head -12 $O/trace_gaps.txt
