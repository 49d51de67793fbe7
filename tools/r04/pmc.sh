#!/bin/bash
# PMC passes for one command (run on the GPU box): one rocprofv3 invocation per counter set, --kernel-trace only.
#   tools/r03/pmc.sh <out dir under gpurun_out> <kernel-name filter (regex)> <command...>
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=$PWD/gpurun_out/$1; TAG=$(echo $1 | tr "/" "_"); FILTER=$2; shift 2; mkdir -p $OUT
CMD="$@"
export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  ( cd /tmp && timeout 40 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_${TAG}_$i -- $CMD > /tmp/pmc_${TAG}_$i.log 2>&1 )
  f=$(ls /tmp/pmc_${TAG}_$i/*/*counter_collection.csv 2>/dev/null | head -1)
  if [ -n "$f" ]; then cp $f $OUT/set$i.csv; else echo "set $i failed"; tail -5 /tmp/pmc_${TAG}_$i.log; fi
done
FILTER="$FILTER" OUTDIR="$OUT" python - <<'PY'
import csv, glob, collections, os, re
out = os.environ['OUTDIR']; flt = re.compile(os.environ['FILTER'])
for f in sorted(glob.glob(out + '/set*.csv')):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if flt.search(k):
            acc[k[:60]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, d in acc.items():
        print(k, {c: '%.5g (n=%d)' % (v[-1], len(v)) for c, v in d.items()})
PY
