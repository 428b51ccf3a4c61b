#!/bin/bash
# The evidence of a round in one GPU call: bench lines (config 3 with HBM traffic measured in-run, configs 2 and 4), rocprofv3 kernel trace of the bench command,
# SQ counters of the fast kernel (two --pmc passes, kernel trace only), phase cycles at three workgroups per CU and alone, one rank's work of the item-sharded
# pipelines at G = 2 / 4 / 8, sessions of up to 5 / 8 / 10 / 20 items, single-query latency.  Raw output under gpurun_out/<tag>p/; the summaries there named
# <tag>_* are what gets copied to profiles/.
# usage (on the GPU box): bash tools/profiles.sh r05
set -x
tag=${1:-rXX}
R=$PWD; O=$R/gpurun_out/${tag}p; mkdir -p $O
python bench.py > $O/${tag}_bench_cfg3.json 2> $O/bench_cfg3.err
python bench.py --config cfg2 --no-cpu-baseline --mode replicas --no-measure-traffic > $O/${tag}_bench_cfg2.json 2>/dev/null
python bench.py --config cfg4 --no-cpu-baseline --mode replicas --no-measure-traffic > $O/${tag}_bench_cfg4.json 2>/dev/null
python tools/latency_probe.py cfg3 > $O/${tag}_latency_cfg3.txt 2>&1
SRN_HOST_CHUNKS=1 python tools/phase_profile.py cfg3 131072 > $O/phase_cfg3.log 2>&1
SRN_HOST_CHUNKS=1 python tools/phase_profile.py cfg3 300 > $O/phase_cfg3_alone.log 2>&1
(echo "# tools/phase_profile.py cfg3 131072 (three workgroups per CU):"; cat $O/phase_cfg3.log; echo; echo "# tools/phase_profile.py cfg3 300 (a query alone on its CU):"; cat $O/phase_cfg3_alone.log) > $O/${tag}_phase_cycles_cfg3.txt
(for G in 2 4 8; do SRN_NB_PHASES=1 python tools/shard_rank_time.py cfg3 $G 2>&1 | tail -5; done) > $O/${tag}_shard_rank_time.txt 2>&1
(for mi in 5 8 10 20; do python tools/long_sessions_bench.py cfg3 262144 $mi 2>&1 | tail -4; done) > $O/${tag}_long_sessions.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python $R/bench.py --steps 10 --warmup 3 --no-sweep --no-cpu-baseline --mode replicas --no-measure-traffic --no-local-g8 > $O/kt.log 2>&1
export SRN_HOST_CHUNKS=1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace -d $O/pmc_sq_a -o pmc --output-format csv -- python $R/tools/count_run.py cfg3 131072 > $O/pmc_sq_a.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS --kernel-trace -d $O/pmc_sq_b -o pmc --output-format csv -- python $R/tools/count_run.py cfg3 131072 > $O/pmc_sq_b.log 2>&1
unset SRN_HOST_CHUNKS
cd $R
python tools/rocprof_summarize.py kernel_trace $O/kt > $O/${tag}_kernel_trace_cfg3.txt
python tools/rocprof_summarize.py sq $O/pmc_sq_a $O/pmc_sq_b 131072 $O/phase_cfg3.log > $O/${tag}_sq_counters_cfg3.json
rm -rf $O/kt $O/pmc_sq_a $O/pmc_sq_b
tail -c 900 $O/${tag}_bench_cfg3.json; head -14 $O/${tag}_kernel_trace_cfg3.txt; grep -A14 derived $O/${tag}_sq_counters_cfg3.json; cat $O/${tag}_shard_rank_time.txt; cat $O/${tag}_long_sessions.txt
