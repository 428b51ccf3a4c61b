#!/bin/bash
# single-query latency (tools/latency_probe.py) with the latency path's launches in one phase (SRN_TINY_PHASES=0, round 4), finish-big kept in the first (1) and finish alone (2)
# -> gpurun_out/latency_ab.txt
mkdir -p gpurun_out; out=gpurun_out/latency_ab.txt; : > $out
for rep in 1 2; do for ph in 0 1 2; do
  echo "== SRN_TINY_PHASES=$ph" >> $out; SRN_TINY_PHASES=$ph python tools/latency_probe.py cfg3 2>&1 | grep "^cfg3" >> $out
done; done
cat $out
