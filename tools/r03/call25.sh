#!/bin/bash
# GPU call 25: PMC counters of the plane conv kernel on the 12 trunk launches of a bench step
set -u
R=$PWD
timeout 200 bash tools/r03/pmc.sh r03_c25/pmc_conv "conv3x3_kernel" $R/tools/_bin/pl_check $R/neural-motifs_amd/csrc/libmotifs_hip.so --conv-replay 2>&1 | tail -4 | cut -c1-600
ls gpurun_out/r03_c25/pmc_conv
