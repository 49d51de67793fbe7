#!/bin/bash
# GPU call 13 (round 5): where the host spends the SGDet step (cfg3 is bound by the host's enqueue time)
set -u
OUT=gpurun_out/r05_c13; mkdir -p $OUT
timeout 300 python tools/r05/host_profile.py cfg3 10 > $OUT/host_profile_cfg3.txt 2>&1
grep -n "img/s\|function calls" $OUT/host_profile_cfg3.txt | head -5 | cut -c1-300
timeout 200 python bench.py --config cfg3 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_cfg3.json
python -c "
import json; d=json.loads(open('$OUT/bench_cfg3.json').read()); print('cfg3', round(d['value'],1), d['ms_per_step'], d.get('step_ms',{}).get('host_p50'))"
