set -x
mkdir -p gpurun_out/final
python bench.py > gpurun_out/final/bench_cfg3.json 2> gpurun_out/final/bench_cfg3.err
python bench.py --config cfg2 --no-cpu-baseline > gpurun_out/final/bench_cfg2.json 2>/dev/null
python bench.py --config cfg4 --no-cpu-baseline > gpurun_out/final/bench_cfg4.json 2>/dev/null
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/final/kt -o kt -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/final/kt.log 2>&1
cd $R
find gpurun_out/final/kt -name "*.db" | head
python profiles/summarize_rocpd.py $(find gpurun_out/final/kt -name "*.db" | head -1) > gpurun_out/final/kernel_trace.txt
bash tools/pmc_bench.sh fetch FETCH_SIZE
bash tools/pmc_bench.sh write WRITE_SIZE
python tools/pmc_traffic.py gpurun_out/pmcb_fetch gpurun_out/pmcb_write cfg3 131072 > gpurun_out/final/traffic_cfg3.json
rm -rf gpurun_out/final/kt gpurun_out/pmcb_fetch gpurun_out/pmcb_write
tail -c 1500 gpurun_out/final/bench_cfg3.json
