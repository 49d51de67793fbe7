#!/bin/bash
# round-3 GPU call 1: first run of the plane GEMM engine (pl_gemm.hip): accuracy + speed through the C ABI, the op-level
# GPU tests on the re-routed mh_gemm_f32, a short bench
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r03_c1; mkdir -p $O
export TMPDIR=/tmp
( timeout 500 tools/_bin/pl_check neural-motifs_amd/csrc/libmotifs_hip.so ) > $O/pl_check.jsonl 2>&1
echo "== pl_check rc=$?"; grep -c '"ok": true' $O/pl_check.jsonl; grep '"ok": false\|error\|summary' $O/pl_check.jsonl | head -20
grep speed $O/pl_check.jsonl | cut -c1-260
( timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x 2>&1 ) > $O/test_gpu_ops.log 2>&1
echo "== test_gpu_ops: $(grep -E ' passed| failed|Aborted|Memory access fault' $O/test_gpu_ops.log | tail -2 | tr '\n' ' ')"
grep -E "^FAILED|^E   " $O/test_gpu_ops.log | head -12 | cut -c1-300
( timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline ) > $O/bench.log 2>&1
echo "== bench: $(tail -1 $O/bench.log | cut -c1-600)"
