#!/bin/bash
R=$PWD; O=$R/gpurun_out/r03; mkdir -p $O
timeout 600 python tools/host_pipe_probe.py cfg3 > $O/host_pipe_probe3.txt 2>&1
grep -v "^\[srn\]" $O/host_pipe_probe3.txt
timeout 900 python -m pytest tests/test_gpu_shard_group.py tests/test_gpu_sharded.py -x -q > $O/pytest5.log 2>&1; echo "pytest rc=$?" >> $O/pytest5.log; tail -5 $O/pytest5.log
timeout 900 python bench.py --mode item-sharded --steps 10 --no-cpu-baseline > $O/bench_item_sharded_g1_b.json 2> $O/bench_item_sharded_g1_b.err; python - <<'PY'
import json
r=json.load(open("gpurun_out/r03/bench_item_sharded_g1_b.json")); print("item-sharded G=1: %.2f M q/s, %.3f ms per %d"%(r["value"]/1e6, r["ms_per_step"], r["config"]["batch"]), r["exchange_bytes_per_query_rank0"])
PY
