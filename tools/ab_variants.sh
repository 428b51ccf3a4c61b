#!/bin/bash
# times the variant libraries given as arguments (names under serenade_amd/variants/) with tools/fast_time.py, twice each, alternating
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for v in "$@"; do SRN_LIB_PATH=$PWD/serenade_amd/variants/libserenade_hip_$v.so python tools/fast_time.py cfg3 2>&1 | tail -1; done; done
