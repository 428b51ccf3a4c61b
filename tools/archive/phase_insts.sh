#!/bin/bash
# Instructions per phase: the kernel built with -DSRN_STOP_AT=N leaves every query after phase tick N; the differences of the
# SQ instruction counters between consecutive builds are the phases' own counts.  Needs serenade_amd/lib_stop_<N>.so.
R=$PWD
cd /tmp && export TMPDIR=/tmp
cp $R/serenade_amd/libserenade_hip.so /tmp/lib_full.so
for k in 0 1 2 3 4 8 9 10 full; do
  if [ $k = full ]; then cp /tmp/lib_full.so $R/serenade_amd/libserenade_hip.so; else cp $R/serenade_amd/lib_stop_$k.so $R/serenade_amd/libserenade_hip.so; fi
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --kernel-trace -d $R/gpurun_out/pi_$k -o pmc --output-format csv -- python $R/tools/count_run.py cfg3 32768 > $R/gpurun_out/pi_$k.log 2>&1
  echo "== stop at $k: $(grep 'main ms' $R/gpurun_out/pi_$k.log)"
  python $R/tools/pmc_sum.py $R/gpurun_out/pi_$k | grep -v "4 dispatches\|3 dispatches" ; python $R/tools/pmc_sum.py $R/gpurun_out/pi_$k
  rm -rf $R/gpurun_out/pi_$k
done
