#!/bin/bash
R=$PWD; O=$R/gpurun_out/r03; mkdir -p $O
python tools/fast_time.py cfg3 > $O/fast_ab1.txt 2>&1
SRN_LIB_PATH=$R/serenade_amd/variants/libserenade_hip_norep.so python tools/fast_time.py cfg3 >> $O/fast_ab1.txt 2>&1
python tools/fast_time.py cfg3 >> $O/fast_ab1.txt 2>&1
SRN_LIB_PATH=$R/serenade_amd/variants/libserenade_hip_norep.so python tools/fast_time.py cfg3 >> $O/fast_ab1.txt 2>&1
grep "fast kernel" $O/fast_ab1.txt
timeout 600 python tools/host_pipe_probe.py cfg3 > $O/host_pipe_probe6.txt 2>&1
grep -v "^\[srn\]" $O/host_pipe_probe6.txt | grep -v "chunks="
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3
