#!/bin/bash
# mid-size device-resident batches under rocprofv3 --kernel-trace
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/batch_trace; mkdir -p $O
cd $R
python tools/batch_trace.py 1,16,256,4096 40 > $O/plain.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python tools/batch_trace.py 1,16,256,4096 40 > $O/traced.txt 2>&1
python tools/batch_trace.py --analyze $O/kt > $O/analysis.txt 2>&1
cat $O/plain.txt | tail -4; cat $O/analysis.txt
rm -rf $O/kt
