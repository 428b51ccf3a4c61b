#!/bin/bash
# Memory-path counters of the item shard's wave-per-query back end (vmis_shard_back_kernel) over the batches of tools/shard_rank_time.py: L2 hits / misses / fabric reads,
# L1 -> L2 read requests and their latency, texture-addresser busy / stalled cycles (own rocprofv3 --pmc passes, kernel trace only).
# usage: tools/pmc_sback_mem.sh <tag> [config] [G]   -> gpurun_out/pmc_<tag>_mem.txt
R=$PWD; tag=$1; cfg=${2:-cfg3}; G=${3:-8}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_sum --kernel-trace -d $R/gpurun_out/pmcm_a -o pmc --output-format csv -- python $R/tools/shard_rank_time.py $cfg $G > $R/gpurun_out/pmcm_a.log 2>&1
rocprofv3 --pmc TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE --kernel-trace -d $R/gpurun_out/pmcm_b -o pmc --output-format csv -- python $R/tools/shard_rank_time.py $cfg $G > $R/gpurun_out/pmcm_b.log 2>&1
(python $R/tools/pmc_sum.py $R/gpurun_out/pmcm_a; python $R/tools/pmc_sum.py $R/gpurun_out/pmcm_b) | grep "^sback" > $R/gpurun_out/pmc_${tag}_mem.txt
rm -rf $R/gpurun_out/pmcm_a $R/gpurun_out/pmcm_b
cat $R/gpurun_out/pmc_${tag}_mem.txt; tail -2 $R/gpurun_out/pmcm_a.log | cut -c1-200
