R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --kernel-trace -d $R/gpurun_out/pf_a -o pmc --output-format csv -- python $R/tools/count_run.py cfg3 32768 > $R/gpurun_out/pf_a.log 2>&1
python $R/tools/pmc_sum.py $R/gpurun_out/pf_a
rocprofv3 --pmc SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD --kernel-trace -d $R/gpurun_out/pf_b -o pmc --output-format csv -- python $R/tools/count_run.py cfg3 32768 > $R/gpurun_out/pf_b.log 2>&1
python $R/tools/pmc_sum.py $R/gpurun_out/pf_b
rm -rf $R/gpurun_out/pf_a $R/gpurun_out/pf_b
