#!/bin/bash
# Round-3 evidence in one GPU call: the whole -m gpu suite (with config 5), bench lines (configs 3, 2, 4, item-sharded group at N=1), kernel trace, HBM traffic
# (own PMC passes), SQ counters, phase cycles, latency, serving, host-pointer sweep.  Raw output under gpurun_out/r03p/, summaries are copied to profiles/ afterwards.
set -x
R=$PWD; O=$R/gpurun_out/r03p; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q -s > $O/pytest_full.log 2>&1; echo "pytest rc=$?" >> $O/pytest_full.log; tail -4 $O/pytest_full.log
python bench.py > $O/bench_cfg3.json 2> $O/bench_cfg3.err
python bench.py --config cfg2 --no-cpu-baseline > $O/bench_cfg2.json 2>/dev/null
python bench.py --config cfg4 --no-cpu-baseline > $O/bench_cfg4.json 2>/dev/null
python bench.py --mode item-sharded --steps 10 > $O/bench_item_sharded_g1_cfg3.json 2>/dev/null
python tools/latency_probe.py cfg3 > $O/latency_cfg3.txt 2>&1
SRN_HOST_CHUNKS=1 python tools/phase_profile.py cfg3 131072 > $O/phase_cfg3.log 2>&1
SRN_SERVE_LANES=0,4 SRN_SERVE_SECONDS=3 python tools/serve_bench.py cfg3 > $O/serving_cfg3.json 2> $O/serving_cfg3.err
SRN_HOST_TRACE=1 python tools/host_pipe_probe.py cfg3 > $O/host_pipe_probe.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python $R/bench.py --steps 10 --warmup 3 --no-sweep --no-cpu-baseline > $O/kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch -o pmc --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-sweep --no-cpu-baseline --parity 0 > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write -o pmc --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-sweep --no-cpu-baseline --parity 0 > $O/pmc_write.log 2>&1
export SRN_HOST_CHUNKS=1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace -d $O/pmc_sq_a -o pmc --output-format csv -- python $R/tools/count_run.py cfg3 131072 > $O/pmc_sq_a.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS --kernel-trace -d $O/pmc_sq_b -o pmc --output-format csv -- python $R/tools/count_run.py cfg3 131072 > $O/pmc_sq_b.log 2>&1
unset SRN_HOST_CHUNKS
cd $R
python tools/r02_summarize.py kernel_trace $O/kt > $O/r03_kernel_trace_cfg3.txt
python tools/r02_summarize.py traffic $O/pmc_fetch $O/pmc_write cfg3 1048576 > $O/r03_traffic_cfg3.json
python tools/r02_summarize.py sq $O/pmc_sq_a $O/pmc_sq_b 131072 $O/phase_cfg3.log > $O/r03_sq_counters_cfg3.json
rm -rf $O/kt/*/ $O/pmc_fetch $O/pmc_write $O/pmc_sq_a $O/pmc_sq_b
tail -c 800 $O/bench_cfg3.json; cat $O/r03_kernel_trace_cfg3.txt; head -c 1200 $O/r03_traffic_cfg3.json; grep -A12 derived $O/r03_sq_counters_cfg3.json
