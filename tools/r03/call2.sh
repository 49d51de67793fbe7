#!/bin/bash
# round-3 GPU call 2: Linear layers on cached plane images; gradient parity at each tensor's own scale with forced kink
# decisions; whole GPU suite; bench + kernel stats; PMC of the plane GEMM
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r03_c2; mkdir -p $O
export TMPDIR=/tmp
for n in test_gpu_ops test_gpu_model test_gpu_configs test_gpu_sgdet test_gpu_baselines test_gpu_dist; do
  ( timeout 900 python -m pytest tests/$n.py -m gpu -q -s 2>&1 ) > $O/$n.log 2>&1
  echo "== $n: $(grep -E ' passed| failed|Aborted|Memory access fault' $O/$n.log | tail -2 | tr '\n' ' ')"
  grep -E "^FAILED|^E   " $O/$n.log | head -12 | cut -c1-300
done
grep -h "kink site\|rows beyond" $O/*.log | head -60
( timeout 300 python bench.py --steps 30 --warmup 6 --no-cpu-baseline ) > $O/bench.log 2>&1
echo "== bench: $(tail -1 $O/bench.log | cut -c1-900)"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $OLDPWD/bench.py --steps 8 --warmup 4 --no-cpu-baseline > $OLDPWD/$O/prof_bench.log 2>&1 )
cp $(ls /tmp/prof/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv 2>/dev/null
head -30 $O/kernel_stats.csv | cut -c1-200
timeout 500 bash tools/r03/pmc.sh r03_c2/pmc_gemm "gemm_kernel" tools/_bin/pl_check neural-motifs_amd/csrc/libmotifs_hip.so --pmc 2>&1 | tail -12 | cut -c1-1200
