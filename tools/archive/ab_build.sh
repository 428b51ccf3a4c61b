#!/bin/bash
# A/B of srn_fast.hip: builds serenade_amd/variants/libserenade_hip_base.so from the COMMITTED srn_fast.hip (git HEAD) and the regular library from the
# working tree; tools/ab_run.sh then times both in one GPU call (tools/fast_time.py: fast-kernel ms + a checksum of the results).
set -e
F=serenade_amd/csrc/srn_fast.hip
cp $F /tmp/srn_fast_new.hip
git show HEAD:$F > $F
python -c "from serenade_amd import build as b; print(b.build_variant('base', []))"
cp /tmp/srn_fast_new.hip $F
touch $F
python -c "from serenade_amd import build as b; b.build_all(verbose=False)"
