#!/bin/bash
# Round-2 evidence in one GPU call: bench lines (configs 3, 2, 4), kernel trace, HBM traffic (own PMC passes), SQ counters, phase cycles,
# sharded capacity mode, latency.  Raw output under gpurun_out/r02/, summaries are copied to profiles/ by hand afterwards.
set -x
R=$PWD; O=$R/gpurun_out/r02; mkdir -p $O
python bench.py > $O/bench_cfg3.json 2> $O/bench_cfg3.err
python bench.py --config cfg2 --no-cpu-baseline > $O/bench_cfg2.json 2>/dev/null
python bench.py --config cfg4 --no-cpu-baseline > $O/bench_cfg4.json 2>/dev/null
python bench.py --mode item-sharded --steps 10 > $O/bench_sharded_g1_cfg3.json 2>/dev/null
python bench.py --mode item-sharded --shard-pipeline stages --steps 10 --no-cpu-baseline > $O/bench_sharded_stages_g1_cfg3.json 2>/dev/null
( python tools/shard_lists_rank_time.py cfg3 8; python tools/shard_lists_rank_time.py cfg3 2; python tools/shard_lists_rank_time.py cfg5_8th 8 65536 ) 2>&1 | grep -v "^  File\|Error\|Exception ignored\|Traceback\|amdgpu.ids" > $O/shard_lists_rank_time.txt
python tools/latency_probe.py cfg3 > $O/latency_cfg3.txt 2>&1
python tools/phase_profile.py cfg3 131072 > $O/phase_cfg3.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python $R/bench.py --steps 10 --warmup 3 --no-sweep --no-cpu-baseline > $O/kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch -o pmc --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-sweep --no-cpu-baseline --parity 0 > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write -o pmc --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-sweep --no-cpu-baseline --parity 0 > $O/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace -d $O/pmc_sq_a -o pmc --output-format csv -- python $R/tools/count_run.py cfg3 131072 > $O/pmc_sq_a.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS --kernel-trace -d $O/pmc_sq_b -o pmc --output-format csv -- python $R/tools/count_run.py cfg3 131072 > $O/pmc_sq_b.log 2>&1
rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVES_EQ_64 SQ_INST_LEVEL_LDS --kernel-trace -d $O/pmc_sq_c -o pmc --output-format csv -- python $R/tools/count_run.py cfg3 131072 > $O/pmc_sq_c.log 2>&1
cd $R
python tools/r02_summarize.py kernel_trace $O/kt > $O/r02_kernel_trace_cfg3.txt
python tools/r02_summarize.py traffic $O/pmc_fetch $O/pmc_write cfg3 1048576 > $O/r02_traffic_cfg3.json
python tools/r02_summarize.py sq $O/pmc_sq_a $O/pmc_sq_b 131072 $O/phase_cfg3.log > $O/r02_sq_counters_cfg3.json
python tools/pmc_sum.py $O/pmc_sq_c > $O/pmc_sq_c.txt
rm -rf $O/kt/*/ $O/pmc_fetch $O/pmc_write $O/pmc_sq_a $O/pmc_sq_b $O/pmc_sq_c
tail -c 600 $O/bench_cfg3.json; cat $O/r02_kernel_trace_cfg3.txt; head -c 1500 $O/r02_traffic_cfg3.json
