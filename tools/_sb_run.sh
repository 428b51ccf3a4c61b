mkdir -p gpurun_out; rm -f gpurun_out/sb_time.log
for knob in none SRN_SBACK_STREAM none; do
  echo "== $knob" >> gpurun_out/sb_time.log
  (env $knob=1 SRN_NB_PHASES=1 timeout 600 python tools/shard_rank_time.py cfg3 8 2>&1 | tail -4) >> gpurun_out/sb_time.log
done
cat gpurun_out/sb_time.log
