#!/bin/bash
R=$PWD; O=$R/gpurun_out/r03; mkdir -p $O
SRN_HOST_TRACE=1 timeout 900 python bench.py --steps 5 --no-cpu-baseline > $O/bench_trace.json 2> $O/bench_trace.err; grep "host pipe: nq 1048576" $O/bench_trace.err | tail -6; grep "host pipe: nq 65536" $O/bench_trace.err | tail -3
python - <<'PY'
import json
r=[json.loads(l) for l in open("gpurun_out/r03/bench_trace.json") if l.startswith("{")][0]
print("value %.2f"%(r["value"]/1e6)); 
for s in r["latency"]["batch_sweep"]: print(s["batch"], "%.3f %.3f"%(s["device_resident"]["ms_p50"], s["host_inclusive"]["ms_p50"]))
PY
timeout 900 python bench.py --mode item-sharded --steps 10 --no-cpu-baseline > $O/bench_item_sharded_g1_c.json 2> $O/bench_item_sharded_g1_c.err; tail -3 $O/bench_item_sharded_g1_c.err; head -c 400 $O/bench_item_sharded_g1_c.json
