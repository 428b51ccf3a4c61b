#!/bin/bash
# round 3, GPU call 2: the shard group's tests, the sharded suite, the host-pipe probe, the serving sweep with CPU / throttle counters
R=$PWD; O=$R/gpurun_out/r03; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_shard_group.py tests/test_gpu_sharded.py -x -q > $O/pytest2.log 2>&1; echo "pytest rc=$?" >> $O/pytest2.log
tail -30 $O/pytest2.log
timeout 600 python tools/host_pipe_probe.py cfg3 > $O/host_pipe_probe.txt 2>&1
cat $O/host_pipe_probe.txt | tail -80
SRN_SERVE_LANES=4 SRN_SERVE_SECONDS=3 timeout 600 python tools/serve_bench.py cfg3 > $O/serving2_cfg3.json 2> $O/serving2_cfg3.err
grep requests_per_s $O/serving2_cfg3.err | cut -c1-420
