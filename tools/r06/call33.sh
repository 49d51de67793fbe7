#!/bin/bash
# GPU call 33 (round 6): cfg3 (SGDet training, detector stage two batches ahead on a worker thread) at run-ahead bound 1 against 4, four rounds
set -u
OUT=gpurun_out/r06_c33; mkdir -p $OUT
for rep in 1 2 3 4; do for a in 1 4; do MOTIFS_MAX_AHEAD=$a timeout 400 python bench.py --config cfg3 --steps 16 --warmup 4 2>/dev/null | tail -1 > $OUT/cfg3_a${a}_$rep.json; python -c "
import json; d=json.loads(open('$OUT/cfg3_a${a}_$rep.json').read()); print('cfg3 ahead$a', round(d['value'],1), d['unit'], round(d['ms_per_step'],2), 'ms')"; done; done
