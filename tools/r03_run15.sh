#!/bin/bash
R=$PWD; O=$R/gpurun_out/r03; mkdir -p $O; V=$R/serenade_amd/variants
: > $O/fast_ab3.txt
for v in default mg6 mg12 mg16 mg24 default; do
  if [ $v = default ]; then python tools/fast_time.py cfg3 >> $O/fast_ab3.txt 2>&1; else SRN_LIB_PATH=$V/libserenade_hip_$v.so python tools/fast_time.py cfg3 >> $O/fast_ab3.txt 2>&1; fi
done
grep "fast kernel" $O/fast_ab3.txt
