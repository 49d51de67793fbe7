#!/bin/bash
# round-2 GPU call 2: (1) activation-plane conv kernels: correctness, per-layer A/B; (2) the f16x3 build through the
# WHOLE -m gpu suite (incl. the BASELINE-config parity tests); (3) bf16x6 default through the config tests; (4) bench
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r02_c2; mkdir -p $O
export TMPDIR=/tmp
F16=$PWD/neural-motifs_amd/csrc/_variants/f16x3/libmotifs_hip.so
( timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -s -k "planes or stem or lstm_barrier or packed_conv" 2>&1 | tail -70 ) > $O/pytest_planes.log 2>&1
tail -30 $O/pytest_planes.log
timeout 300 python tools/gpu_perf_conv.py all > $O/perf_default.log 2>&1
MH_PCONV_TILE=128 timeout 300 python tools/gpu_perf_conv.py conv > $O/perf_tile128.log 2>&1
MH_PCONV_TILE=256 timeout 300 python tools/gpu_perf_conv.py conv > $O/perf_tile256.log 2>&1
MH_GEMM_PATCH=rows timeout 300 python tools/gpu_perf_conv.py gemm > $O/perf_gemm_rows.log 2>&1
MOTIFS_HIP_LIB=$F16 timeout 300 python tools/gpu_perf_conv.py all > $O/perf_f16x3.log 2>&1
for f in perf_default perf_tile128 perf_tile256 perf_gemm_rows perf_f16x3; do echo "== $f"; grep -v "^{" $O/$f.log | grep -v amdgpu.ids; done
( MOTIFS_HIP_LIB=$F16 timeout 1200 python -m pytest tests -m gpu -q -s 2>&1 ) > $O/pytest_f16x3_full.log 2>&1
echo "== f16x3 full suite"; grep -E "^cfg|passed|failed|FAILED|Error" $O/pytest_f16x3_full.log | tail -60
( timeout 900 python -m pytest tests/test_gpu_configs.py -m gpu -q -s 2>&1 ) > $O/pytest_cfg_bf16x6.log 2>&1
echo "== bf16x6 config tests"; grep -E "^cfg|passed|failed|FAILED|Error" $O/pytest_cfg_bf16x6.log | tail -40
( timeout 500 python bench.py --steps 20 --warmup 5 ) > $O/bench.log 2>&1
tail -1 $O/bench.log | cut -c1-2500
( MOTIFS_HIP_LIB=$F16 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline ) > $O/bench_f16x3.log 2>&1
tail -1 $O/bench_f16x3.log | cut -c1-900
