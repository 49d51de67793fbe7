#!/bin/bash
# GPU call 8: what the matrix-core stem waits for (no stores / no loads / no epilogue; blocks per CU)
LIB=$PWD/neural-motifs_amd/csrc/libmotifs_hip.so
for d in 0 1 2 3 4 7; do echo "debug $d"; MH_STEM_DEBUG=$d tools/_bin/pl_check $LIB --stem; done
for b in 2 4 16 64; do echo "blocks/CU $b"; MH_STEM_BLOCKS=$b tools/_bin/pl_check $LIB --stem; done
