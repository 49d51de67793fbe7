#!/bin/bash
# GPU call 16 (round 6): tower weight gradient with the mask fragments shared through LDS: tests, kernel time per variant (channels per
# wave x row ranges), bench
set -u
OUT=gpurun_out/r06_c16; mkdir -p $OUT; R=$PWD
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "tower" > $OUT/tests_tower.log 2>&1; grep -E "passed|failed|rror" $OUT/tests_tower.log | tail -3 | cut -c1-200
cat > /tmp/t1.py <<'PY'
import os, sys, torch
sys.path[:0] = [os.environ['R'], os.path.join(os.environ['R'], 'neural-motifs_amd')]
from lib import _hip as hip
N, C0 = 1536, 256
rects = torch.rand(N, 27, 27, 2).cuda(); dy = torch.randn(N, 14, 14, C0).cuda()
xp = hip.tower_conv1_pad(rects)
for _ in range(5): hip.tower_conv1_wgrad(xp, dy)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20): hip.tower_conv1_wgrad(xp, dy)
e.record(); torch.cuda.synchronize()
print('NB', os.environ.get('MH_TOWER_WGRAD_NB', '2'), 'blocks', os.environ.get('MH_TOWER_WGRAD_BLOCKS', '256'), 'us per call (kernel + reduce)', round(s.elapsed_time(e) / 20 * 1e3, 1))
PY
for nb in 2 1; do for bl in 256 512; do R=$R MH_TOWER_WGRAD_NB=$nb MH_TOWER_WGRAD_BLOCKS=$bl python /tmp/t1.py 2>&1 | tail -1; done; done
show() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], round(d['value'],1), d['unit'], 'ms', round(d['ms_per_step'],2), 'p50', d.get('ms_per_step_p50'), 'cal', round(d['calibration']['plane_gemm_4096_tflops']),
          'seg', {k: round(v, 2) for k, v in d['main_stream_segments'].items() if k.endswith('_ms')})
except Exception as e:
    print(sys.argv[1], 'unreadable', e)
PY
}
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline"
timeout 200 $B > $OUT/bench_new.json 2> $OUT/bench_new.err; show $OUT/bench_new.json
timeout 200 $B > $OUT/bench_new_b.json 2> /dev/null; show $OUT/bench_new_b.json
