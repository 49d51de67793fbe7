#!/bin/bash
# GPU call 19 (round 5): fabric-side traffic of the step's big matrix products (fc6 forward / input gradient / weight gradient at
# 1536 rows, the 120-row object fc6, fc7) on the shipped library: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes
# over tools/_bin/gemm_traffic (tools/traffic_run.sh gemm); summarised offline into profiles/r05_gemm_traffic_summary.json
set -u
bash tools/traffic_run.sh gemm > gpurun_out/traffic_run.log 2>&1
mkdir -p gpurun_out/r05_c19; cp gpurun_out/traffic/gemm.* gpurun_out/r05_c19/ 2>/dev/null; ls -la gpurun_out/r05_c19 | cut -c1-120; tail -3 gpurun_out/traffic_run.log | cut -c1-200
