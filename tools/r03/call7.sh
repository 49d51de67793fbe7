#!/bin/bash
# round-3 GPU call 7: diagnosis of the varying cfg1 relation-logit error; small products back on the in-loop kernel;
# new parity tests (cfg3 / cfg5 sizes, cfg4 model); bench + trace gaps
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=gpurun_out/r03_c7; mkdir -p $O
export TMPDIR=/tmp
( timeout 400 python tools/r03/diag_cfg1.py ) > $O/diag_cfg1.log 2>&1; grep "^img" $O/diag_cfg1.log | cut -c1-200; tail -3 $O/diag_cfg1.log | cut -c1-300
( timeout 900 python -m pytest tests/test_gpu_sgdet.py -m gpu -q -s 2>&1 ) > $O/test_gpu_sgdet.log 2>&1
echo "== test_gpu_sgdet: $(grep -E ' passed| failed|Aborted|Memory access fault' $O/test_gpu_sgdet.log | tail -2 | tr '\n' ' ')"
grep -E "^FAILED|^E   " $O/test_gpu_sgdet.log | head -12 | cut -c1-300
grep "cfg3\|cfg5" $O/test_gpu_sgdet.log | grep -v grad | head -20 | cut -c1-200
( timeout 600 python -m pytest tests/test_gpu_configs.py -m gpu -q -s -k cfg4 2>&1 ) > $O/test_cfg4.log 2>&1
echo "== cfg4: $(grep -E ' passed| failed|Aborted|Memory access fault' $O/test_cfg4.log | tail -2 | tr '\n' ' ')"; grep -E "^FAILED|^E   |cfg4" $O/test_cfg4.log | head -14 | cut -c1-250
( timeout 300 python bench.py --steps 30 --warmup 6 --no-cpu-baseline ) > $O/bench.log 2>&1
echo "== bench: $(tail -1 $O/bench.log | cut -c1-400)"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $R/bench.py --steps 8 --warmup 4 --no-cpu-baseline > $R/$O/prof_bench.log 2>&1 )
cp $(ls /tmp/prof/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv 2>/dev/null
cp $(ls /tmp/prof/*/*kernel_trace.csv | head -1) $O/kernel_trace.csv 2>/dev/null
python tools/trace_gaps.py $O/kernel_trace.csv --steps 3 --top 14 > $O/trace_gaps.txt 2>&1; head -22 $O/trace_gaps.txt | cut -c1-200
( timeout 200 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --host-profile ) > $O/bench_host.log 2> $O/host_profile.txt; grep "host enqueue" $O/host_profile.txt | cut -c1-300
