#!/bin/bash
# host-pointer batches of 16 384 .. 65 536 queries: one chunk (default) against two chunks with the first SRN_HOST_FIRST_PCT % of the batch -> gpurun_out/host_cut_ab.txt
mkdir -p gpurun_out; out=gpurun_out/host_cut_ab.txt; : > $out
for pct in 0 50 62 75 87; do
  echo "== SRN_HOST_FIRST_PCT=$pct" >> $out
  SRN_HOST_FIRST_PCT=$pct timeout 600 python tools/host_batch_floor.py cfg3 16384 32768 65536 2>&1 | grep "^cfg3" | cut -c1-110 >> $out
done
cat $out
