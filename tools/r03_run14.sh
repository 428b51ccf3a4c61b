#!/bin/bash
R=$PWD; O=$R/gpurun_out/r03; mkdir -p $O
timeout 600 python tools/host_pipe_probe.py cfg3 > $O/host_pipe_probe7.txt 2>&1
grep -v "^\[srn\]" $O/host_pipe_probe7.txt | grep -v "download kernel"
grep "nq 65536" $O/host_pipe_probe7.txt | awk 'NR%9==2' | head -8
