#!/bin/bash
# GPU call 9: bisect of the two-stream race + cfg4 (BatchNorm statistics at C = 2048)
set -u
OUT=gpurun_out/r03_c9; mkdir -p $OUT
run() { tag=$1; shift; env "$@" timeout 240 python tools/r03/diag3_streams.py "$tag" 2>&1 | grep -E "VARIANT|Error|error" | tail -3 >> $OUT/diag3.log; }
run default            A=1
run trunk_v2           MOTIFS_TRUNK=v2
run trunk_fp32_epi     MOTIFS_TRUNK_DIRECT=0
run linear_inloop      MOTIFS_LINEAR=inloop
run v2_and_inloop      MOTIFS_TRUNK=v2 MOTIFS_LINEAR=inloop
run no_caching_alloc   PYTORCH_NO_CUDA_MEMORY_CACHING=1
run serialize_kernels  AMD_SERIALIZE_KERNEL=3
cat $OUT/diag3.log
timeout 600 python -m pytest tests/test_gpu_configs.py -x -q -k "cfg4" 2>&1 | tail -15 > $OUT/cfg4_test.log
tail -5 $OUT/cfg4_test.log
timeout 300 python bench.py --config cfg4 --steps 10 --warmup 3 2>&1 | tail -3 > $OUT/bench_cfg4.log
tail -2 $OUT/bench_cfg4.log
bash tools/r03/traffic.sh r03_c9 2>&1 | tail -3
