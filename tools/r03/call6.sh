#!/bin/bash
# round-3 GPU call 6: image-output epilogue (conv -> next layer's plane image, no fp32 round trip) + stem to image: checks, model
# parity, cfg1 error survey on both trunks, bench with the new meters
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=gpurun_out/r03_c6; mkdir -p $O
export TMPDIR=/tmp
( timeout 600 tools/_bin/pl_check neural-motifs_amd/csrc/libmotifs_hip.so --conv ) > $O/pl_conv_check.jsonl 2>&1
echo "== pl_check conv rc=$?"; grep -c '"ok": true' $O/pl_conv_check.jsonl; grep '"ok": false\|error\|summary' $O/pl_conv_check.jsonl | head -20 | cut -c1-400
grep "image output" $O/pl_conv_check.jsonl | cut -c1-260 | head -24
for n in test_gpu_model test_gpu_configs; do
  ( timeout 900 python -m pytest tests/$n.py -m gpu -q -s 2>&1 ) > $O/$n.log 2>&1
  echo "== $n: $(grep -E ' passed| failed|Aborted|Memory access fault' $O/$n.log | tail -2 | tr '\n' ' ')"
  grep -E "^FAILED|^E   " $O/$n.log | head -12 | cut -c1-300
done
grep -h "trunk feature map" $O/*.log | head
( MOTIFS_TRUNK=planes timeout 300 python tools/r03/cfg1_errors.py ) > $O/cfg1_planes.log 2>&1; grep "cfg1 img" $O/cfg1_planes.log
( MOTIFS_TRUNK=planes MOTIFS_TRUNK_DIRECT=0 timeout 300 python tools/r03/cfg1_errors.py ) > $O/cfg1_planes_nodirect.log 2>&1; grep "cfg1 img" $O/cfg1_planes_nodirect.log
( MOTIFS_TRUNK=v2 timeout 300 python tools/r03/cfg1_errors.py ) > $O/cfg1_v2.log 2>&1; grep "cfg1 img" $O/cfg1_v2.log
( timeout 300 python bench.py --steps 30 --warmup 6 --no-cpu-baseline ) > $O/bench.log 2>&1
echo "== bench (plane trunk, direct): $(tail -1 $O/bench.log | cut -c1-3000)"
( MOTIFS_TRUNK_DIRECT=0 timeout 300 python bench.py --steps 30 --warmup 6 --no-cpu-baseline ) > $O/bench_nodirect.log 2>&1
echo "== bench (plane trunk, converters): $(tail -1 $O/bench_nodirect.log | cut -c1-200)"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $R/bench.py --steps 8 --warmup 4 --no-cpu-baseline > $R/$O/prof_bench.log 2>&1 )
cp $(ls /tmp/prof/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv 2>/dev/null
head -30 $O/kernel_stats.csv | cut -c1-150
