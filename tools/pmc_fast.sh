#!/bin/bash
# SQ counters of the predict kernels over three full launches of tools/count_run.py (own rocprofv3 --pmc passes, kernel trace only).
# usage: tools/pmc_fast.sh <tag> [config] [batch]   -> gpurun_out/pmc_<tag>_{a,b}.txt
R=$PWD; tag=$1; cfg=${2:-cfg3}; B=${3:-32768}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace -d $R/gpurun_out/pmcf_a -o pmc --output-format csv -- python $R/tools/count_run.py $cfg $B > $R/gpurun_out/pmcf_a.log 2>&1
python $R/tools/pmc_sum.py $R/gpurun_out/pmcf_a > $R/gpurun_out/pmc_${tag}_a.txt
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS --kernel-trace -d $R/gpurun_out/pmcf_b -o pmc --output-format csv -- python $R/tools/count_run.py $cfg $B > $R/gpurun_out/pmcf_b.log 2>&1
python $R/tools/pmc_sum.py $R/gpurun_out/pmcf_b > $R/gpurun_out/pmc_${tag}_b.txt
rm -rf $R/gpurun_out/pmcf_a $R/gpurun_out/pmcf_b
cat $R/gpurun_out/pmc_${tag}_a.txt $R/gpurun_out/pmc_${tag}_b.txt
