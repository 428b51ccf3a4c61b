#!/bin/bash
# round 3, GPU call 4: config 5 test, host-pipe probe after the slot rework
R=$PWD; O=$R/gpurun_out/r03; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_config5.py -x -q -s > $O/pytest_cfg5.log 2>&1; echo "pytest rc=$?" >> $O/pytest_cfg5.log
tail -15 $O/pytest_cfg5.log
timeout 600 python tools/host_pipe_probe.py cfg3 > $O/host_pipe_probe2.txt 2>&1
grep -v "^\[srn\]" $O/host_pipe_probe2.txt
