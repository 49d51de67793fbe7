#!/bin/bash
# GPU call 11 (round 4): side-stream priority x backward order A/B (unprofiled bench, alternating), new tests
set -u
OUT=gpurun_out/r04_c11; mkdir -p $OUT
for rep in 1 2; do for prio in 0 -1; do for late in 0 auto; do
  MOTIFS_SIDE_PRIORITY=$prio MOTIFS_LATE_VR=$late timeout 100 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --h2d-steps 0 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('prio=$prio late=$late', round(d['value'],1), 'img/s p50', d['ms_per_step_p50'], 'max', d['ms_per_step_max'], 'lstm fwd/bwd us', round(d['hbm_kernels']['lstm_fwd']['us_per_call']), round(d['hbm_kernels']['lstm_bwd']['us_per_call']))"
done; done; done
timeout 600 python -m pytest tests/test_gpu_dist.py tests/test_gpu_sgdet.py -x -q -s > $OUT/tests.log 2>&1; grep -E "passed|failed" $OUT/tests.log | tail -1; grep -E "^cfg5|^sgdet e2e|R@20" $OUT/tests.log | cut -c1-230
