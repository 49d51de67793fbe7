#!/bin/bash
# PMC passes for the two MFMA kernels (run on the GPU box): one rocprofv3 invocation per counter set
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc; mkdir -p $OUT
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc$i -- python $GRAFT_REPO_ROOT/tools/pmc_mfma.py > /tmp/pmc$i.log 2>&1
  f=$(ls /tmp/pmc$i/*/*counter_collection.csv 2>/dev/null | head -1)
  if [ -n "$f" ]; then cp $f $OUT/set$i.csv; else tail -5 /tmp/pmc$i.log; fi
done
python - <<'PY'
import csv, glob, collections, os
out = os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/pmc'
for f in sorted(glob.glob(out + '/set*.csv')):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:40]
        if 'gemm_kernel' in k or 'conv3x3' in k:
            acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, d in acc.items():
        print(k, {c: '%.4g' % (sum(v[-1:]) ) for c, v in d.items()})
PY
