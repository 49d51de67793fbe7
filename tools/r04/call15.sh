#!/bin/bash
# GPU call 15 (round 4): forward LSTM layers on tagged granules: LSTM / decoder tests, model tests, A/B of the bench
set -u
OUT=gpurun_out/r04_c15; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "lstm or decoder or hwcell" > $OUT/lstm_tests.log 2>&1; tail -3 $OUT/lstm_tests.log | cut -c1-300
timeout 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_baselines.py -x -q > $OUT/model_tests.log 2>&1; tail -3 $OUT/model_tests.log | cut -c1-300
for rep in 1 2; do for g in 0 1; do
  MH_LSTM_GRAN=$g timeout 100 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --h2d-steps 0 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('gran=$g', round(d['value'],1), 'img/s mean', round(d['ms_per_step'],2), 'p50', d['ms_per_step_p50'], 'lstm fwd/bwd us', round(d['hbm_kernels']['lstm_fwd']['us_per_call']), round(d['hbm_kernels']['lstm_bwd']['us_per_call']))"
done; done
for g in 0 1; do MH_LSTM_GRAN=$g timeout 200 python bench.py --config cfg3 --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('cfg3 gran=$g', round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms')"; done
