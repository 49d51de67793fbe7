#!/bin/bash
# round-3 GPU call 8: which intermediate differs between two-stream and single-stream evaluation; RoIAlign with chunk-fastest
# block order; new op tests
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=gpurun_out/r03_c8; mkdir -p $O
export TMPDIR=/tmp
( timeout 300 python tools/r03/diag2_cfg1.py ) > $O/diag2.log 2>&1; grep "^img" $O/diag2.log | cut -c1-700; tail -2 $O/diag2.log | cut -c1-300
( timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -s 2>&1 ) > $O/test_gpu_ops.log 2>&1
echo "== test_gpu_ops: $(grep -E ' passed| failed|Aborted|Memory access fault' $O/test_gpu_ops.log | tail -2 | tr '\n' ' ')"
grep -E "^FAILED|^E   |trunk image" $O/test_gpu_ops.log | head -16 | cut -c1-300
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline > $R/$O/prof_bench.log 2>&1 )
cp $(ls /tmp/prof/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv 2>/dev/null
grep "roi_align\|act_planes\|stem_kernel" $O/kernel_stats.csv | cut -c1-200
tail -1 $O/prof_bench.log | cut -c1-300
