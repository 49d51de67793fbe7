#!/bin/bash
# round-2 GPU call 5: fused exponent passes -- correctness (ops, sgdet, model), perf, traffic summaries (conv + GEMM), bench, profile
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r02_c5; mkdir -p $O
export TMPDIR=/tmp
for n in test_gpu_ops test_gpu_sgdet test_gpu_model; do
  ( timeout 900 python -m pytest tests/$n.py -m gpu -q -s 2>&1 ) > $O/$n.log 2>&1
  echo "== $n: $(grep -E ' passed| failed|Aborted|Memory access fault' $O/$n.log | tail -2 | tr '\n' ' ')"
  grep -E "^FAILED|^E   " $O/$n.log | head -12
done
timeout 200 python tools/gpu_perf_conv.py all > $O/perf_default.log 2>&1
echo "== perf_default"; grep -v "^{" $O/perf_default.log | grep -E "TRUNK|GEMM|conv" | cut -c1-60
bash tools/traffic_run.sh conv > $O/traffic_conv.log 2>&1
bash tools/traffic_run.sh gemm default=MH_NOP=1 rows=MH_GEMM_PATCH=rows > $O/traffic_gemm.log 2>&1
cp gpurun_out/traffic/*.csv gpurun_out/traffic/*.jsonl $O/ 2>/dev/null; ls $O | tr '\n' ' '
( timeout 500 python bench.py --steps 20 --warmup 5 ) > $O/bench.log 2>&1
echo "== bench"; tail -1 $O/bench.log | cut -c1-1200
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $OLDPWD/bench.py --steps 8 --warmup 3 --no-cpu-baseline > $OLDPWD/$O/prof_bench.log 2>&1
cd $OLDPWD; cp $(ls /tmp/prof/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv 2>/dev/null; head -16 $O/kernel_stats.csv | cut -c1-140
