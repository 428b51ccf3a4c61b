#!/bin/bash
# Round-4 evidence in one GPU call: bench lines (config 3 with traffic measured in-run, configs 2 and 4, item-sharded alone), kernel trace, SQ counters, phase cycles at
# three workgroups per CU and alone, per-phase instruction counts and stop-build durations (needs tools/fast_phase_build.sh run in the CPU container first), rank times
# of the sharded pipelines, latency / serving / host-pointer sweeps.  Raw output under gpurun_out/r04p/, summaries are copied to profiles/ afterwards.
set -x
R=$PWD; O=$R/gpurun_out/r04p; mkdir -p $O
python bench.py --measure-traffic > $O/bench_cfg3.json 2> $O/bench_cfg3.err
python bench.py --config cfg2 --no-cpu-baseline --mode replicas > $O/bench_cfg2.json 2>/dev/null
python bench.py --config cfg4 --no-cpu-baseline --mode replicas > $O/bench_cfg4.json 2>/dev/null
python tools/latency_probe.py cfg3 > $O/latency_cfg3.txt 2>&1
SRN_HOST_CHUNKS=1 python tools/phase_profile.py cfg3 131072 > $O/phase_cfg3.log 2>&1
SRN_HOST_CHUNKS=1 python tools/phase_profile.py cfg3 300 > $O/phase_cfg3_alone.log 2>&1
(for G in 2 4 8; do python tools/shard_rank_time.py cfg3 $G 2>&1 | tail -3; done) > $O/shard_rank_time.txt 2>&1
SRN_SERVE_LANES=0,4 SRN_SERVE_SECONDS=3 python tools/serve_bench.py cfg3 > $O/serving_cfg3.json 2> $O/serving_cfg3.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python $R/bench.py --steps 10 --warmup 3 --no-sweep --no-cpu-baseline --mode replicas > $O/kt.log 2>&1
export SRN_HOST_CHUNKS=1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace -d $O/pmc_sq_a -o pmc --output-format csv -- python $R/tools/count_run.py cfg3 131072 > $O/pmc_sq_a.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS --kernel-trace -d $O/pmc_sq_b -o pmc --output-format csv -- python $R/tools/count_run.py cfg3 131072 > $O/pmc_sq_b.log 2>&1
unset SRN_HOST_CHUNKS
cd $R
python tools/r02_summarize.py kernel_trace $O/kt > $O/r04_kernel_trace_cfg3.txt
python tools/r02_summarize.py sq $O/pmc_sq_a $O/pmc_sq_b 131072 $O/phase_cfg3.log > $O/r04_sq_counters_cfg3.json
rm -rf $O/kt/*/ $O/pmc_sq_a $O/pmc_sq_b
if [ -f serenade_amd/lib_stop_full.so.bin ]; then
  cp serenade_amd/libserenade_hip.so /tmp/libserenade_hip.keep
  bash tools/fast_phase_insts.sh > /dev/null 2>&1
  cp gpurun_out/fast_phase_insts.txt $O/r04_fast_phase_insts_raw.txt
  cp /tmp/libserenade_hip.keep serenade_amd/libserenade_hip.so
fi
tail -c 600 $O/bench_cfg3.json; cat $O/r04_kernel_trace_cfg3.txt | head -12; grep -A12 derived $O/r04_sq_counters_cfg3.json; cat $O/shard_rank_time.txt
