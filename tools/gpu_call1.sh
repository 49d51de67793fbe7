#!/bin/bash
# round-2 GPU call 1: full -m gpu suite (incl. the BASELINE-config parity tests), bench, schedule A/B, f16x3 evaluation
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r02_c1; mkdir -p $O
export TMPDIR=/tmp
nproc > $O/nproc.txt
( time timeout 1500 python -m pytest tests -m gpu -x -q -s 2>&1 ) > $O/pytest.log 2>&1
tail -5 $O/pytest.log
( time timeout 600 python bench.py --steps 20 --warmup 5 ) > $O/bench.log 2>&1
tail -3 $O/bench.log | cut -c1-1500
( MH_CONV_SCHEDULE=uniform timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline ) > $O/bench_uniform.log 2>&1
tail -1 $O/bench_uniform.log | cut -c1-400
timeout 200 python tools/gpu_perf_mfma.py > $O/perf_default.log 2>&1
MH_CONV_SCHEDULE=uniform timeout 200 python tools/gpu_perf_mfma.py > $O/perf_uniform.log 2>&1
MOTIFS_HIP_LIB=$PWD/neural-motifs_amd/csrc/_variants/f16x3/libmotifs_hip.so timeout 200 python tools/gpu_perf_mfma.py > $O/perf_f16x3.log 2>&1
timeout 300 tools/_bin/split_check neural-motifs_amd/csrc/libmotifs_hip.so neural-motifs_amd/csrc/_variants/f16x3/libmotifs_hip.so > $O/split_check.jsonl 2> $O/split_check.err
( MOTIFS_HIP_LIB=$PWD/neural-motifs_amd/csrc/_variants/f16x3/libmotifs_hip.so timeout 400 python -m pytest tests/test_gpu_ops.py -m gpu -q 2>&1 | tail -40 ) > $O/pytest_f16x3_ops.log 2>&1
( MOTIFS_HIP_LIB=$PWD/neural-motifs_amd/csrc/_variants/f16x3/libmotifs_hip.so timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline ) > $O/bench_f16x3.log 2>&1
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $OLDPWD/bench.py --steps 8 --warmup 3 --no-cpu-baseline > $OLDPWD/$O/prof_bench.log 2>&1
cd $OLDPWD; cp $(ls /tmp/prof/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv 2>/dev/null
for f in perf_default perf_uniform perf_f16x3; do echo "== $f"; cat $O/$f.log; done
echo "== split_check"; cut -c1-300 $O/split_check.jsonl | head -40
echo "== f16x3 ops"; tail -15 $O/pytest_f16x3_ops.log
echo "== bench f16x3"; tail -1 $O/bench_f16x3.log | cut -c1-600
