#!/bin/bash
# round-2 GPU call 12: optimizer chunk table expanded on the device (no 140 KB copy before the clip kernels) -- is the stall gone?
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r02_c12; mkdir -p $O
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "sgd or clip or optim" 2>&1 ) > $O/test_optim.log 2>&1; tail -2 $O/test_optim.log
( timeout 300 python bench.py --steps 30 --warmup 6 --no-cpu-baseline --host-profile ) > $O/bench.log 2> $O/probe.txt
echo "== bench: $(tail -1 $O/bench.log | cut -c1-160)"; grep -E "^probe|host enqueue" $O/probe.txt | cut -c1-700
( HSA_ENABLE_SDMA=0 timeout 300 python bench.py --steps 30 --warmup 6 --no-cpu-baseline --host-profile ) > $O/bench_nosdma.log 2> $O/probe_nosdma.txt
echo "== bench HSA_ENABLE_SDMA=0: $(tail -1 $O/bench_nosdma.log | cut -c1-160)"; grep -E "^probe|host enqueue" $O/probe_nosdma.txt | cut -c1-700
( MOTIFS_MAX_AHEAD=-1 timeout 300 python bench.py --steps 30 --warmup 6 --no-cpu-baseline ) > $O/bench_nothrottle.log 2>&1
echo "== bench, no run-ahead bound: $(tail -1 $O/bench_nothrottle.log | cut -c1-160)"
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $OLDPWD/bench.py --steps 8 --warmup 4 --no-cpu-baseline > $OLDPWD/$O/prof_bench.log 2>&1
cd $OLDPWD; cp $(ls /tmp/prof/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv 2>/dev/null
cp $(ls /tmp/prof/*/*kernel_trace.csv | head -1) $O/kernel_trace.csv 2>/dev/null
python tools/trace_gaps.py $O/kernel_trace.csv --steps 3 --top 25 > $O/trace_gaps.txt 2>&1; head -8 $O/trace_gaps.txt
