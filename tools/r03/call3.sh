#!/bin/bash
# round-3 GPU call 3: first run of the plane conv (pl_conv.hip): accuracy vs float64 / the round-2 kernel + per-layer speed;
# the tests touched since call 2 (reference-kernel goldens, fused-SGD guard, detector pre-training with forced kinks, the
# trunk on the plane engine inside the model); bench with both trunks; PMC of the plane GEMM and the plane conv
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=gpurun_out/r03_c3; mkdir -p $O
export TMPDIR=/tmp
( timeout 600 tools/_bin/pl_check neural-motifs_amd/csrc/libmotifs_hip.so --conv --speed ) > $O/pl_conv_check.jsonl 2>&1
echo "== pl_check conv rc=$?"; grep -c '"ok": true' $O/pl_conv_check.jsonl; grep '"ok": false\|error\|summary' $O/pl_conv_check.jsonl | head -20 | cut -c1-400
grep "conv speed" $O/pl_conv_check.jsonl | cut -c1-200
for n in test_gpu_ops test_gpu_model test_gpu_sgdet test_gpu_configs; do
  ( timeout 900 python -m pytest tests/$n.py -m gpu -q -s 2>&1 ) > $O/$n.log 2>&1
  echo "== $n: $(grep -E ' passed| failed|Aborted|Memory access fault' $O/$n.log | tail -2 | tr '\n' ' ')"
  grep -E "^FAILED|^E   " $O/$n.log | head -12 | cut -c1-300
done
( timeout 300 python bench.py --steps 30 --warmup 6 --no-cpu-baseline ) > $O/bench.log 2>&1
echo "== bench (plane trunk): $(tail -1 $O/bench.log | cut -c1-700)"
( MOTIFS_TRUNK=v2 timeout 300 python bench.py --steps 30 --warmup 6 --no-cpu-baseline ) > $O/bench_v2trunk.log 2>&1
echo "== bench (v2 trunk): $(tail -1 $O/bench_v2trunk.log | cut -c1-300)"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $R/bench.py --steps 8 --warmup 4 --no-cpu-baseline > $R/$O/prof_bench.log 2>&1 )
cp $(ls /tmp/prof/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv 2>/dev/null
head -16 $O/kernel_stats.csv | cut -c1-160
timeout 400 bash tools/r03/pmc.sh r03_c3/pmc_gemm "gemm_kernel" $R/tools/_bin/pl_check $R/neural-motifs_amd/csrc/libmotifs_hip.so --pmc 2>&1 | tail -8 | cut -c1-1500
