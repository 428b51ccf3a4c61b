#!/bin/bash
# Round 4, second half (MID instantiation, latency path): the bench lines and the kernel trace at the final build in one GPU call.  Raw output under gpurun_out/r04q/.
set -x
R=$PWD; O=$R/gpurun_out/r04q; mkdir -p $O
( time python bench.py --measure-traffic > $O/bench_cfg3.json 2> $O/bench_cfg3.err ) 2> $O/bench_cfg3.time
python bench.py --config cfg2 --no-cpu-baseline --mode replicas > $O/bench_cfg2.json 2>/dev/null
python bench.py --config cfg4 --no-cpu-baseline --mode replicas > $O/bench_cfg4.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python $R/bench.py --steps 10 --warmup 3 --no-sweep --no-cpu-baseline --mode replicas > $O/kt.log 2>&1
cd $R
python tools/r02_summarize.py kernel_trace $O/kt > $O/r04_kernel_trace_cfg3.txt
rm -rf $O/kt/*/
tail -c 400 $O/bench_cfg3.json; cat $O/bench_cfg3.time; head -14 $O/r04_kernel_trace_cfg3.txt
