#!/bin/bash
# round-2 GPU call 16 (last): host-side pair enumeration in GT-box evaluation, pinned Blobs -- model parity + cfg1 / default bench
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r02_c16; mkdir -p $O
export TMPDIR=/tmp
( timeout 400 python -m pytest tests/test_gpu_model.py -m gpu -q -s 2>&1 ) > $O/test_gpu_model.log 2>&1
echo "== test_gpu_model: $(grep -E ' passed| failed|Aborted|Memory access fault' $O/test_gpu_model.log | tail -2 | tr '\n' ' ')"
grep -E "^FAILED|^E   " $O/test_gpu_model.log | head -12 | cut -c1-300
( timeout 100 python bench.py --config cfg1 --no-cpu-baseline ) > $O/bench_cfg1.log 2>&1
echo "== cfg1: $(tail -1 $O/bench_cfg1.log | cut -c1-200)"
( timeout 150 python bench.py --steps 30 --warmup 6 --no-cpu-baseline ) > $O/bench.log 2>&1
echo "== bench: $(tail -1 $O/bench.log | cut -c1-200)"
