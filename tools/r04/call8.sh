#!/bin/bash
# GPU call 8 (round 4): sgdet train parity with / without the ring shapes; cfg4 at its stated size
set -u
OUT=gpurun_out/r04_c8; mkdir -p $OUT
for ring in 0 1; do
  MH_PL_RING=$ring timeout 300 python -m pytest tests/test_gpu_sgdet.py -x -q -s -k "test_sgdet_train_step_parity" > $OUT/sgdet_ring$ring.log 2>&1
  echo "ring=$ring"; grep -E "passed|failed" $OUT/sgdet_ring$ring.log | tail -1; grep -E "sgdet grad" $OUT/sgdet_ring$ring.log | sort -t= -k4 | awk '{print}' | sort -k10 -g | tail -4 | cut -c1-170
done
timeout 900 python -m pytest tests/test_gpu_configs.py -x -q -s -k "test_cfg4_resnet_sgcls_train_step_b6_1536_rows" > $OUT/cfg4_full.log 2>&1
grep -E "passed|failed|Error" $OUT/cfg4_full.log | tail -3 | cut -c1-300; grep -E "^cfg4|^kink" $OUT/cfg4_full.log | cut -c1-170 | head -150
