#!/bin/bash
R=$PWD; O=$R/gpurun_out/r03; mkdir -p $O; V=$R/serenade_amd/variants
: > $O/fast_ab2.txt
for v in default nodefer O3 default O2 licm ilp unroll nodefer; do
  if [ $v = default ]; then python tools/fast_time.py cfg3 >> $O/fast_ab2.txt 2>&1; else SRN_LIB_PATH=$V/libserenade_hip_$v.so python tools/fast_time.py cfg3 >> $O/fast_ab2.txt 2>&1; fi
done
grep "fast kernel" $O/fast_ab2.txt
timeout 600 python tools/phase_profile.py cfg3 131072 > $O/phase_cfg3_r03a.txt 2>&1; tail -32 $O/phase_cfg3_r03a.txt | head -24
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "not config4 and not baseline_configs" 2>&1 | tail -3
