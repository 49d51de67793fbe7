#!/bin/bash
# GPU call 13 (round 4): frozen-trunk prefetch: equality test, then A/B of the bench (serial / background stream / CU-masked)
set -u
OUT=gpurun_out/r04_c13; mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_model.py -x -q -k "prefetched_trunk" 2>&1 | tail -5 | cut -c1-300
for rep in 1 2; do
for arm in "--no-prefetch:0" ":0" ":32" ":64" ":128" ":192"; do
  flag=${arm%%:*}; cus=${arm##*:}
  MOTIFS_PREFETCH_CUS=$cus timeout 100 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --h2d-steps 4 $flag 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); t=d['roofline']['trunk_only']; print('arm [$flag] cus=$cus:', round(d['value'],1), 'img/s mean', round(d['ms_per_step'],2), 'p50', d['ms_per_step_p50'], 'max', d['ms_per_step_max'], 'h2d', round(d['h2d_inclusive']['ms_per_step'],2), 'trunk ms', round(t['ms_per_step'],2), 'lstm f/b', round(d['hbm_kernels']['lstm_fwd']['us_per_call']), round(d['hbm_kernels']['lstm_bwd']['us_per_call']))"
done; done
