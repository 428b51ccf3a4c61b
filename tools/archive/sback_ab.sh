#!/bin/bash
# A/B of the item shard's wave-per-query back end in ONE GPU call: one rank's time at G = 8 on config 3 (tools/shard_rank_time.py) for each setting, then the back end's
# oracle tests.  A setting is a library variant (serenade_amd/variants/libserenade_hip_<name>.so, built by serenade_amd.build.build_variant) or NAME=VALUE (environment knob).
# usage: bash tools/sback_ab.sh [<name> | <KNOB>=<value> ...]   -> gpurun_out/sback_ab.txt
mkdir -p gpurun_out; out=gpurun_out/sback_ab.txt; : > $out
for v in "$@" default; do
  unset SRN_LIB_PATH; envs=""
  case "$v" in default) ;; *=*) envs="$v" ;; *) export SRN_LIB_PATH=$PWD/serenade_amd/variants/libserenade_hip_$v.so ;; esac
  echo "== $v" >> $out
  env $envs SRN_NB_PHASES=1 timeout 900 python tools/shard_rank_time.py ${CFG:-cfg3} ${G:-8} 2>&1 | tail -5 | grep -v "LISTS pipeline" >> $out
done
unset SRN_LIB_PATH
timeout 1500 python -m pytest tests/test_gpu_shard_group.py -x -q 2>&1 | tail -5 >> $out
cat $out
