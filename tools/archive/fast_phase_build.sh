#!/bin/bash
# builds serenade_amd/lib_stop_<N>.so.bin for tools/fast_phase_insts.sh (CPU container): only srn_fast.hip is recompiled per variant
set -e
O=serenade_amd/csrc/_obj
python -c "from serenade_amd import build as b; b.build_all(verbose=False)"
OBJS=$(ls $O/*.o | grep -v srn_fast)
for k in 0 1 3 2 4 8 9 10 11 12 full; do
  D=""; [ $k != full ] && D="-DSRN_FAST_STOP=$k"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -Os -std=c++17 -fPIC -Wall -Wno-unused-value -mllvm -disable-machine-licm $D -c -o /tmp/srn_fast_$k.o serenade_amd/csrc/srn_fast.hip
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o serenade_amd/lib_stop_$k.so.bin $OBJS /tmp/srn_fast_$k.o
done
