#!/bin/bash
# SQ counters of the item shard's wave-per-query back end (vmis_shard_back_kernel) over the batches of tools/shard_rank_time.py (own rocprofv3 --pmc passes, kernel trace only).
# usage: tools/pmc_sback.sh <tag> [config] [G]   -> gpurun_out/pmc_<tag>_{a,b}.txt
R=$PWD; tag=$1; cfg=${2:-cfg3}; G=${3:-8}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace -d $R/gpurun_out/pmcs_a -o pmc --output-format csv -- python $R/tools/shard_rank_time.py $cfg $G > $R/gpurun_out/pmcs_a.log 2>&1
python $R/tools/pmc_sum.py $R/gpurun_out/pmcs_a > $R/gpurun_out/pmc_${tag}_a.txt
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS --kernel-trace -d $R/gpurun_out/pmcs_b -o pmc --output-format csv -- python $R/tools/shard_rank_time.py $cfg $G > $R/gpurun_out/pmcs_b.log 2>&1
python $R/tools/pmc_sum.py $R/gpurun_out/pmcs_b > $R/gpurun_out/pmc_${tag}_b.txt
rm -rf $R/gpurun_out/pmcs_a $R/gpurun_out/pmcs_b
grep "^sback\|^fast" $R/gpurun_out/pmc_${tag}_a.txt $R/gpurun_out/pmc_${tag}_b.txt
