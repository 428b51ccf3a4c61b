#!/bin/bash
# Instructions per phase of vmis_fast_kernel: libraries built with -DSRN_FAST_STOP=N leave every query after phase tick N; the
# differences of the SQ instruction counters between consecutive builds are the phases' own counts.
# Needs serenade_amd/lib_stop_<N>.so.bin (tools/fast_phase_build.sh).  Output: gpurun_out/fast_phase_insts.txt
R=$PWD
cd /tmp && export TMPDIR=/tmp
out=$R/gpurun_out/fast_phase_insts.txt; : > $out
for k in 0 1 3 2 4 8 9 10 11 12 full; do
  cp $R/serenade_amd/lib_stop_$k.so.bin $R/serenade_amd/libserenade_hip.so
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --kernel-trace -d $R/gpurun_out/pi_$k -o pmc --output-format csv -- python $R/tools/count_run.py cfg3 32768 > $R/gpurun_out/pi_$k.log 2>&1
  echo "== stop at $k: $(grep 'main ms' $R/gpurun_out/pi_$k.log)" >> $out
  python $R/tools/pmc_sum.py $R/gpurun_out/pi_$k | grep "^fast" >> $out
  rm -rf $R/gpurun_out/pi_$k
done
cat $out
