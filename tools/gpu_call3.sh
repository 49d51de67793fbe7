#!/bin/bash
# round-2 GPU call 3: f16x3-default tree: full suite, LDS-DMA probe, schedule / scheduling-knob A/B, benches, profile
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r02_c3; mkdir -p $O
export TMPDIR=/tmp
V=$PWD/neural-motifs_amd/csrc/_variants
( timeout 1500 python -m pytest tests -m gpu -q -s 2>&1 ) > $O/pytest.log 2>&1
echo "== pytest"; grep -E "passed|failed|FAILED|^E  " $O/pytest.log | tail -30
timeout 60 tools/_bin/dma_probe > $O/dma_probe.jsonl 2>&1; echo "== dma_probe"; cat $O/dma_probe.jsonl
timeout 200 python tools/gpu_perf_conv.py all > $O/perf_default.log 2>&1
MH_SLOTS=512 timeout 200 python tools/gpu_perf_conv.py all > $O/perf_slots512.log 2>&1
MH_CONV_SCHEDULE=uniform timeout 200 python tools/gpu_perf_conv.py conv > $O/perf_uniform.log 2>&1
MOTIFS_HIP_LIB=$V/f16_v4/libmotifs_hip.so timeout 200 python tools/gpu_perf_conv.py all > $O/perf_v4.log 2>&1
MOTIFS_HIP_LIB=$V/f16_v11/libmotifs_hip.so timeout 200 python tools/gpu_perf_conv.py all > $O/perf_v11.log 2>&1
for f in perf_default perf_slots512 perf_uniform perf_v4 perf_v11; do echo "== $f"; grep -v "^{" $O/$f.log | grep -E "TRUNK fp32 |GEMM|conv1_2|conv3_2|conv4_2|conv5" | cut -c1-60; done
( timeout 500 python bench.py --steps 20 --warmup 5 ) > $O/bench.log 2>&1
echo "== bench"; tail -1 $O/bench.log | cut -c1-3000
for c in cfg1 cfg3 cfg5 cfg4; do ( timeout 300 python bench.py --config $c --steps 10 --warmup 3 ) > $O/bench_$c.log 2>&1; echo "== bench $c"; tail -2 $O/bench_$c.log | cut -c1-1500; done
timeout 200 tools/_bin/split_check neural-motifs_amd/csrc/libmotifs_hip.so $V/bf16x6/libmotifs_hip.so > $O/split_check.jsonl 2> $O/split_check.err
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $OLDPWD/bench.py --steps 8 --warmup 3 --no-cpu-baseline > $OLDPWD/$O/prof_bench.log 2>&1
cd $OLDPWD; cp $(ls /tmp/prof/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv 2>/dev/null; head -12 $O/kernel_stats.csv | cut -c1-160
