#!/bin/bash
# A/B of fast-kernel library variants (serenade_amd/variants/libserenade_hip_<name>.so) against the default build in ONE GPU call: tools/fast_time.py per variant, two rounds
# alternating.  usage: bash tools/fast_ab.sh <name> [<name> ...]   -> gpurun_out/fast_ab.txt
mkdir -p gpurun_out; out=gpurun_out/fast_ab.txt; : > $out
for rep in 1 2; do
  for v in "$@" default; do
    if [ "$v" = default ]; then unset SRN_LIB_PATH; else export SRN_LIB_PATH=$PWD/serenade_amd/variants/libserenade_hip_$v.so; fi
    timeout 600 python tools/fast_time.py ${CFG:-cfg3} 2>&1 | tail -1 >> $out
  done
done
unset SRN_LIB_PATH; cat $out
