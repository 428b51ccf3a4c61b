#!/bin/bash
# round 3, GPU call 3: timeline of the chunked host path (rocprofv3 kernel + memory-copy trace), serving sweep with the spin-lock combiner, item-sharded bench at N=1
R=$PWD; O=$R/gpurun_out/r03; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace -d $O/hp_trace -o hp --output-format csv -- python $R/tools/host_pipe_trace.py 1048576 3 > $O/hp_trace.log 2>&1
cd $R
tail -4 $O/hp_trace.log
python tools/trace_overlap.py $O/hp_trace 45 > $O/hp_trace_overlap.txt 2>&1; cat $O/hp_trace_overlap.txt
GPU_MAX_HW_QUEUES=8 python tools/host_pipe_trace.py 1048576 3 2>&1 | tail -3
rm -rf $O/hp_trace
SRN_SERVE_LANES=4 SRN_SERVE_SECONDS=3 timeout 600 python tools/serve_bench.py cfg3 > $O/serving3_cfg3.json 2> $O/serving3_cfg3.err
grep requests_per_s $O/serving3_cfg3.err | cut -c1-130,250-420
timeout 900 python bench.py --mode item-sharded --steps 10 > $O/bench_item_sharded_g1.json 2> $O/bench_item_sharded_g1.err; tail -c 2500 $O/bench_item_sharded_g1.json; tail -5 $O/bench_item_sharded_g1.err
