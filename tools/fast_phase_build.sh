#!/bin/bash
# builds serenade_amd/lib_stop_<N>.so.bin for tools/fast_phase_insts.sh (CPU container)
for k in 1 2 4 8 9 10 11 12; do
  SRN_CFLAGS="-DSRN_FAST_STOP=$k" python -c "from serenade_amd import build as b; b.build_all(verbose=False)" && cp serenade_amd/libserenade_hip.so serenade_amd/lib_stop_$k.so.bin
done
python -c "from serenade_amd import build as b; b.build_all(force=True, verbose=False)" && cp serenade_amd/libserenade_hip.so serenade_amd/lib_stop_full.so.bin
