#!/bin/bash
# Round 4, second half (MID instantiation, latency path): the bench lines and the kernel trace at the final build in one GPU call.  Raw output under gpurun_out/r04q/.
set -x
R=$PWD; O=$R/gpurun_out/r04q; mkdir -p $O
( time python bench.py --measure-traffic > $O/bench_cfg3.json 2> $O/bench_cfg3.err ) 2> $O/bench_cfg3.time
python bench.py --config cfg2 --no-cpu-baseline --mode replicas > $O/bench_cfg2.json 2>/dev/null
python bench.py --config cfg4 --no-cpu-baseline --mode replicas > $O/bench_cfg4.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python $R/bench.py --steps 10 --warmup 3 --no-sweep --no-cpu-baseline --mode replicas > $O/kt.log 2>&1
export SRN_HOST_CHUNKS=1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace -d $O/pmc_sq_a -o pmc --output-format csv -- python $R/tools/count_run.py cfg3 131072 > $O/pmc_sq_a.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS --kernel-trace -d $O/pmc_sq_b -o pmc --output-format csv -- python $R/tools/count_run.py cfg3 131072 > $O/pmc_sq_b.log 2>&1
cd $R
python tools/phase_profile.py cfg3 131072 > $O/phase_cfg3.log 2>&1
python tools/phase_profile.py cfg3 300 > $O/phase_cfg3_alone.log 2>&1
unset SRN_HOST_CHUNKS
python tools/r02_summarize.py kernel_trace $O/kt > $O/r04_kernel_trace_cfg3.txt
python tools/r02_summarize.py sq $O/pmc_sq_a $O/pmc_sq_b 131072 $O/phase_cfg3.log > $O/r04_sq_counters_cfg3.json
rm -rf $O/kt/*/ $O/pmc_sq_a $O/pmc_sq_b
grep -A8 derived $O/r04_sq_counters_cfg3.json; grep "cyc/query" $O/phase_cfg3.log | head -16
tail -c 400 $O/bench_cfg3.json; cat $O/bench_cfg3.time; head -14 $O/r04_kernel_trace_cfg3.txt
