#!/bin/bash
R=$PWD; O=$R/gpurun_out/r03; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace -d $O/hp_trace2 -o hp --output-format csv -- python $R/tools/host_pipe_trace.py 1048576 3 > $O/hp_trace2.log 2>&1
cd $R
tail -3 $O/hp_trace2.log
python tools/trace_overlap.py $O/hp_trace2 38 > $O/hp_trace2_overlap.txt 2>&1; cat $O/hp_trace2_overlap.txt
rm -rf $O/hp_trace2
