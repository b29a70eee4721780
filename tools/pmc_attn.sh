# PMC counters of the attention kernels alone (tools/attn_bench.py as the workload): two passes, summary to stdout
#   gpurun -- 'bash tools/pmc_attn.sh gpurun_out/pa "--bwd --drop 0.1 --stages 2"'
out="$1"; args="$2"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $out
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --output-format csv -d $out/pmc_a -- python3 tools/attn_bench.py $args --iters 2 > $out/a.log 2>&1 && \
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_SALU --output-format csv -d $out/pmc_b -- python3 tools/attn_bench.py $args --iters 2 > $out/b.log 2>&1 && \
python3 tools/pmc_derived.py $out/pmc_a/*/ $out/pmc_b/*/ 'attn_' > $out/summary.txt 2>&1
rc=$?
rm -rf $out/pmc_a $out/pmc_b
cat $out/summary.txt
exit $rc
