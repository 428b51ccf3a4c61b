#!/bin/bash
# usage: tools/pmc_bench.sh <tag> "<counters>"   -- bench.py (2 timed steps) under rocprofv3 --pmc, own pass per counter set
cd /tmp && export TMPDIR=/tmp
tag=$1; shift
rocprofv3 --pmc $1 --kernel-trace -d /root/repo/gpurun_out/pmcb_$tag -o pmc --output-format csv -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /root/repo/gpurun_out/pmcb_$tag.log 2>&1
grep -o '"kernel_ms_avg": [0-9.]*' /root/repo/gpurun_out/pmcb_$tag.log
