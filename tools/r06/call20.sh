#!/bin/bash
# GPU call 20 (round 6): which round-6 change costs cfg1 (PredCls evaluation, one image per step, host-bound) its 7-20 %?  One switch
# at a time on one box, round 5's tree beside them, three rounds
set -u
OUT=gpurun_out/r06_c20; mkdir -p $OUT; R=$PWD
one() { ( cd $1 && env $4 timeout 400 python bench.py --config cfg1 --steps 16 --warmup 4 2>/dev/null | tail -1 ) > $OUT/$3.json; python -c "
import json; d=json.loads(open('$OUT/$3.json').read()); print('%-28s' % '$2', round(d['value'],1), d['unit'], round(d['ms_per_step'],2), 'ms')"; }
for rep in 1 2 3; do
  one $R new new_$rep X=1
  one $R/_ab/r05 r05 r05_$rep X=1
  one $R no_host_geometry nogeom_$rep MOTIFS_HOST_GEOMETRY=0
  one $R no_pair_product nopair_$rep MOTIFS_PAIR_PRODUCT=0
  one $R both_off bothoff_$rep "MOTIFS_HOST_GEOMETRY=0 MOTIFS_PAIR_PRODUCT=0"
  one $R tower_valu towervalu_$rep "MH_TOWER_CONV1=valu"
  one $R all_off alloff_$rep "MOTIFS_HOST_GEOMETRY=0 MOTIFS_PAIR_PRODUCT=0 MH_TOWER_CONV1=valu MH_STEM=valu MH_GEMM_ORDER=0"
done
