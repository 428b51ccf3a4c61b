#!/bin/bash
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  SRN_LIB_PATH=$PWD/serenade_amd/variants/libserenade_hip_base.so python tools/fast_time.py ${1:-cfg3} 2>&1 | tail -1
  python tools/fast_time.py ${1:-cfg3} 2>&1 | tail -1
done
