#!/bin/bash
# The item shard's wave-per-query back end, measured (round 6): the shard-group tests, one rank's time at G = 2, 4, 8 (tools/shard_rank_time.py), the opt-in forms at G = 8,
# and the SQ / memory-path counters of the kernel (tools/pmc_sback.sh, tools/pmc_sback_mem.sh).   usage: bash tools/sback_profile.sh <tag>   -> gpurun_out/sback_<tag>_*.txt
tag=${1:-r06}; mkdir -p gpurun_out
(timeout 1500 python -m pytest tests/test_gpu_shard_group.py -x -q 2>&1 | tail -5) > gpurun_out/sback_${tag}_tests.txt
: > gpurun_out/sback_${tag}_rank_time.txt
for G in 2 4 8; do echo "+ SRN_NB_PHASES=1 python tools/shard_rank_time.py cfg3 $G" >> gpurun_out/sback_${tag}_rank_time.txt; (SRN_NB_PHASES=1 timeout 600 python tools/shard_rank_time.py cfg3 $G 2>&1 | tail -5) >> gpurun_out/sback_${tag}_rank_time.txt; done
for knob in SRN_SBACK_STREAM SRN_SBACK_PBYTES SRN_SBACK_BITMAP; do echo "+ $knob=1 python tools/shard_rank_time.py cfg3 8" >> gpurun_out/sback_${tag}_rank_time.txt; (env $knob=1 timeout 600 python tools/shard_rank_time.py cfg3 8 2>&1 | tail -3) >> gpurun_out/sback_${tag}_rank_time.txt; done
bash tools/pmc_sback.sh $tag cfg3 8 > gpurun_out/sback_${tag}_pmc.log 2>&1
bash tools/pmc_sback_mem.sh $tag cfg3 8 >> gpurun_out/sback_${tag}_pmc.log 2>&1
cat gpurun_out/sback_${tag}_tests.txt; grep "one rank\|back-end launches" gpurun_out/sback_${tag}_rank_time.txt | cut -c1-300
