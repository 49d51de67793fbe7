#!/bin/bash
# GPU call 12 (round 4): PMC sets + fabric traffic of the shipped trunk kernels, full bench line (with the CPU baseline), secondary
# configurations, the whole -m gpu suite
set -u
OUT=gpurun_out/r04_c12; mkdir -p $OUT; R=$PWD
timeout 300 bash tools/r04/pmc.sh r04_c12/pmc_conv "conv3x3" $R/tools/_bin/pl_check $R/neural-motifs_amd/csrc/libmotifs_hip.so --conv-replay 2>&1 | tail -3 | cut -c1-400
timeout 200 bash tools/r04/traffic.sh r04_c12/traffic 2>&1 | tail -2 | cut -c1-300
timeout 300 bash tools/r04/pmc.sh r04_c12/pmc_gemm "gemm_ring" $R/tools/_bin/pl_check $R/neural-motifs_amd/csrc/libmotifs_hip.so --pmc 3 8 2>&1 | tail -2 | cut -c1-400
timeout 400 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; tail -1 $OUT/bench.json | cut -c1-300
for c in cfg1 cfg3 cfg4 cfg5 recipe; do timeout 300 python bench.py --config $c --steps 10 --warmup 3 2>/dev/null | tail -1 > $OUT/bench_$c.json; python -c "
import json; d=json.loads(open('$OUT/bench_$c.json').read()); print('$c', round(d['value'],1), d['unit'], round(d['ms_per_step'],2), 'ms')"; done
timeout 1200 python -m pytest tests/ -x -q -m gpu > $OUT/gpu_tests.log 2>&1; tail -3 $OUT/gpu_tests.log | cut -c1-300
