#!/bin/bash
R=$PWD; O=$R/gpurun_out/r03; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "resident or device_pointer or tiny_config" 2>&1 | tail -3
for f in "" "--no-resident-flag"; do
timeout 600 python bench.py --steps 40 --no-sweep --no-cpu-baseline $f > $O/bench_res.json 2> $O/bench_res.err; python - <<'PY'
import json
r=[json.loads(l) for l in open("gpurun_out/r03/bench_res.json") if l.startswith("{")][0]
print("value %.2f M q/s ms/step %.3f kernel %.3f other %s"%(r["value"]/1e6, r["ms_per_step"], r["roofline"]["kernel_ms_avg"], r["roofline"]["other_launches_ms_avg"]))
PY
done
