#!/bin/bash
# GPU call 4 (round 5): the tests added / changed since call 3 -- SGDet file (e2e probability bound from the float64 floor), the
# ResNet-101 trunk at its stated size against the float64 floor, AlphaDropout masks injected in the SELU RoI head, the small-product
# engine after the f16x3 arm was removed -- and the cfg2 line of this tree
set -u
OUT=gpurun_out/r05_c4; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_sgdet.py -x -q -m gpu -s > $OUT/sgdet_tests.log 2>&1; grep -E "passed|failed|rror|vs the float64" $OUT/sgdet_tests.log | tail -8 | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu -s -k "resnet" > $OUT/resnet_tests.log 2>&1; grep -E "passed|failed|rror|vs the float64|resnet" $OUT/resnet_tests.log | tail -12 | cut -c1-300
timeout 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "gemm or small_product" > $OUT/ops_tests.log 2>&1; grep -E "passed|failed|rror" $OUT/ops_tests.log | tail -3 | cut -c1-300
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_c4/bench.json').read().strip().splitlines()[-1])
print(round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'unmetered', d['unmetered'], 'dominant', d['roofline']['dominant_class'], round(d['roofline']['frac'],3), d['scaling_diagnostics'])
PY
