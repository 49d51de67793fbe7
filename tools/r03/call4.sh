#!/bin/bash
# round-3 GPU call 4: plane conv after the epilogue-atomics fix (per-layer speed), dual-orientation images, RoIAlign rewrite,
# bench + kernel stats on the plane trunk, fabric traffic of the plane GEMM for two tile shapes
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=gpurun_out/r03_c4; mkdir -p $O
export TMPDIR=/tmp
( timeout 600 tools/_bin/pl_check neural-motifs_amd/csrc/libmotifs_hip.so --conv --speed ) > $O/pl_conv_check.jsonl 2>&1
echo "== pl_check conv rc=$?"; grep -c '"ok": true' $O/pl_conv_check.jsonl; grep '"ok": false\|error\|summary' $O/pl_conv_check.jsonl | head -20 | cut -c1-400
grep "conv speed" $O/pl_conv_check.jsonl | cut -c1-200
( timeout 600 tools/_bin/pl_check neural-motifs_amd/csrc/libmotifs_hip.so ) > $O/pl_check.jsonl 2>&1
echo "== pl_check rc=$?"; grep '"ok": false\|error\|summary\|make_planes' $O/pl_check.jsonl | head -20 | cut -c1-300
for n in test_gpu_ops test_gpu_model; do
  ( timeout 900 python -m pytest tests/$n.py -m gpu -q -s 2>&1 ) > $O/$n.log 2>&1
  echo "== $n: $(grep -E ' passed| failed|Aborted|Memory access fault' $O/$n.log | tail -2 | tr '\n' ' ')"
  grep -E "^FAILED|^E   " $O/$n.log | head -12 | cut -c1-300
done
( timeout 300 python bench.py --steps 30 --warmup 6 --no-cpu-baseline ) > $O/bench.log 2>&1
echo "== bench (plane trunk): $(tail -1 $O/bench.log | cut -c1-400)"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $R/bench.py --steps 8 --warmup 4 --no-cpu-baseline > $R/$O/prof_bench.log 2>&1 )
cp $(ls /tmp/prof/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv 2>/dev/null
head -24 $O/kernel_stats.csv | cut -c1-160
for cfg in "0 8" "1 4" "1 2"; do
  set -- $cfg
  ( cd /tmp && timeout 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf_$1_$2 -- $R/tools/_bin/pl_check $R/neural-motifs_amd/csrc/libmotifs_hip.so --pmc $1 $2 > /tmp/pf_$1_$2.log 2>&1 )
  f=$(ls /tmp/pf_$1_$2/*/*counter_collection.csv | head -1); cp $f $O/fetch_shape$1_sk$2.csv
  echo "== FETCH_SIZE shape $1 splitk $2: $(grep gemm_kernel $f | tail -1 | awk -F, '{print $(NF)}')"
done
