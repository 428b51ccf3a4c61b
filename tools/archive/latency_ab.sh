#!/bin/bash
# single-query latency (tools/latency_probe.py): five launches (SRN_TINY_FUSED=0), the fused one-launch form waiting for the stream (SRN_TINY_SPIN=0), and the default
# (fused, items in the kernel arguments, the caller spins on the kernel's last pinned word); two rounds -> gpurun_out/latency_ab.txt; then the tests that walk the latency path
mkdir -p gpurun_out; out=gpurun_out/latency_ab.txt; : > $out
for rep in 1 2; do
  echo "== SRN_TINY_FUSED=0" >> $out; SRN_TINY_FUSED=0 python tools/latency_probe.py ${CFG:-cfg3} 2>&1 | grep "^cfg" | head -1 >> $out
  echo "== SRN_TINY_SPIN=0" >> $out; SRN_TINY_SPIN=0 python tools/latency_probe.py ${CFG:-cfg3} 2>&1 | grep "^cfg" | head -1 >> $out
  echo "== default" >> $out; python tools/latency_probe.py ${CFG:-cfg3} 2>&1 | grep "^cfg" | head -1 >> $out
done
python tools/latency_probe.py ${CFG:-cfg3} 10 2>&1 | grep "cfg\|sessions of" | head -4 >> $out
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_serving.py tests/test_gpu_mid_tier.py -x -q 2>&1 | tail -3 >> $out
cat $out
