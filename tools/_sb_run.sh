mkdir -p gpurun_out; rm -f gpurun_out/sb_time.log
for lib in new subticks new; do
  if [ $lib = new ]; then unset SRN_LIB_PATH; else export SRN_LIB_PATH=/root/repo/serenade_amd/variants/libserenade_hip_$lib.so; fi
  echo "== $lib" >> gpurun_out/sb_time.log
  (SRN_NB_PHASES=1 timeout 600 python tools/shard_rank_time.py cfg3 8 2>&1 | tail -4) >> gpurun_out/sb_time.log
done
cat gpurun_out/sb_time.log
