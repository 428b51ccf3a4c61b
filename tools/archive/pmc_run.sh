#!/bin/bash
# usage: tools/pmc_run.sh <tag> "<counters>" -- runs tools_exp.py under rocprofv3 --pmc (own pass, no tracing domains)
cd /tmp && export TMPDIR=/tmp
tag=$1; shift
rocprofv3 --pmc $1 --kernel-trace -d /root/repo/gpurun_out/pmc_$tag -o pmc --output-format csv -- python /root/repo/tools/phase_profile.py cfg3 32768 > /root/repo/gpurun_out/pmc_$tag.log 2>&1
ls /root/repo/gpurun_out/pmc_$tag | head
