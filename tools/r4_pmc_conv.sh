# PMC passes over tools/conv_bench.py (one model forward x 6): matrix-pipe busy, instruction mix, LDS conflicts of the conv kernels
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P=gpurun_out/r4p; mkdir -p $P
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --output-format csv -d $P/pmc_a -- python3 tools/conv_bench.py > $P/pmc_a.txt 2>&1 && \
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_SALU --output-format csv -d $P/pmc_b -- python3 tools/conv_bench.py > $P/pmc_b.txt 2>&1
echo rc=$?
python3 tools/pmc_derived.py $P/pmc_a/*/ $P/pmc_b/*/ 'spconv_tile|spconv_split' > $P/pmc_conv.txt 2>&1
rm -f $P/pmc_?/*/*_counter_collection.csv $P/pmc_?/*/*_kernel_trace.csv
cat $P/pmc_conv.txt
