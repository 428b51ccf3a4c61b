#!/bin/bash
R=$PWD; O=$R/gpurun_out/r03; mkdir -p $O
timeout 600 python tools/host_pipe_probe.py cfg3 > $O/host_pipe_probe5.txt 2>&1
grep -v "^\[srn\]" $O/host_pipe_probe5.txt | grep -v "slices"; grep "^\[srn\]" $O/host_pipe_probe5.txt | awk 'NR%6==1' | head -12
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace -d $O/hp_trace3 -o hp --output-format csv -- python $R/tools/host_pipe_trace.py 1048576 3 > $O/hp_trace3.log 2>&1
cd $R
python tools/trace_overlap.py $O/hp_trace3 32 > $O/hp_trace3_overlap.txt 2>&1; head -30 $O/hp_trace3_overlap.txt; grep -c . $O/hp_trace3_overlap.txt
rm -rf $O/hp_trace3
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "chunked or sixty or concurrent or device_pointer" 2>&1 | tail -3
