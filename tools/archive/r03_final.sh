#!/bin/bash
# the round's last GPU call: the whole -m gpu suite, the bench line, and the kernel trace of the same command, at the final build
R=$PWD; O=$R/gpurun_out/r03f; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q -s > $O/pytest_full.log 2>&1; echo "pytest rc=$?" >> $O/pytest_full.log; tail -3 $O/pytest_full.log
python bench.py > $O/bench_cfg3.json 2> $O/bench_cfg3.err; cut -c1-300 $O/bench_cfg3.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python $R/bench.py --steps 10 --warmup 3 --no-sweep --no-cpu-baseline > $O/kt.log 2>&1
cd $R
python tools/r02_summarize.py kernel_trace $O/kt > $O/r03_kernel_trace_cfg3.txt; tail -5 $O/r03_kernel_trace_cfg3.txt
rm -rf $O/kt/*/
