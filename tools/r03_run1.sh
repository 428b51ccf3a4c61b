#!/bin/bash
# round 3, GPU call 1: the whole -m gpu suite on the new host paths, the serving sweep (combiner lanes), the bench line with the new batch sweep
R=$PWD; O=$R/gpurun_out/r03; mkdir -p $O
timeout 1100 python -m pytest tests -m gpu -x -q > $O/pytest1.log 2>&1; echo "pytest rc=$?" >> $O/pytest1.log
tail -5 $O/pytest1.log
SRN_SERVE_LANES=0,2,4,8 SRN_SERVE_SECONDS=3 timeout 600 python tools/serve_bench.py cfg3 > $O/serving_cfg3.json 2> $O/serving_cfg3.err
grep -c requests_per_s $O/serving_cfg3.err
timeout 600 python bench.py --steps 20 > $O/bench_cfg3_run1.json 2> $O/bench_cfg3_run1.err
tail -c 3000 $O/bench_cfg3_run1.json
