set -x
export SEG3D_BENCH_IDLE_PROBE=0  # the idle probe (device-side sleeps, extra steps) must not enter the step cut of the trace
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5p
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5p/fwd -- python3 bench.py --mode fwd --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-exact > gpurun_out/r5p/fwd.json 2> gpurun_out/r5p/fwd.err && \
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5p/default -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-fp32-exact > gpurun_out/r5p/default.json 2> gpurun_out/r5p/default.err && \
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r5p/fetch -- python3 bench.py --mode fwd --steps 2 --warmup 1 --no-cpu-baseline --no-fp32-exact > gpurun_out/r5p/fetch.json 2> gpurun_out/r5p/fetch.err && cp gpurun_out/bench_layers.json gpurun_out/r5p/layers_fetch.json && \
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/r5p/write -- python3 bench.py --mode fwd --steps 2 --warmup 1 --no-cpu-baseline --no-fp32-exact > gpurun_out/r5p/write.json 2> gpurun_out/r5p/write.err && \
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --output-format csv -d gpurun_out/r5p/pmc_a -- python3 bench.py --steps 1 --warmup 1 --scenes 1 --no-cpu-baseline --no-fp32-exact > gpurun_out/r5p/pmc_a.json 2> gpurun_out/r5p/pmc_a.err && \
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_SALU --output-format csv -d gpurun_out/r5p/pmc_b -- python3 bench.py --steps 1 --warmup 1 --scenes 1 --no-cpu-baseline --no-fp32-exact > gpurun_out/r5p/pmc_b.json 2> gpurun_out/r5p/pmc_b.err
rc=$?; echo six passes rc=$rc; [ $rc = 0 ] || exit $rc
export SEG3D_WGRAD_STREAM=0 SEG3D_AUX_OVERLAP=0
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5p/one_stream -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-fp32-exact > gpurun_out/r5p/one_stream.json 2> gpurun_out/r5p/one_stream.err
echo seven passes rc=$?
unset SEG3D_WGRAD_STREAM SEG3D_AUX_OVERLAP
# post-process on the box (the raw counter tables exceed what travels back), keep summaries + the kernel traces of the step cuts
P=gpurun_out/r5p; S=$P/summaries; mkdir -p $S
python3 tools/pmc_derived.py $P/pmc_a/*/ $P/pmc_b/*/ 'attn_' > $S/pmc_attention.txt 2>&1
python3 tools/pmc_derived.py $P/pmc_a/*/ $P/pmc_b/*/ 'spconv_tile|spconv_split|wgrad' > $S/pmc_conv.txt 2>&1
python3 tools/pmc_conv_traffic.py $P/fetch/*/*_counter_collection.csv $P/write/*/*_counter_collection.csv $P/layers_fetch.json $S/pmc_conv_traffic.json > $S/pmc_conv_traffic.log 2>&1
python3 tools/pmc_kernel_traffic.py $P/fetch/*/*_counter_collection.csv $P/write/*/*_counter_collection.csv 'attn_fused_fwd' > $S/pmc_attention_traffic.txt 2>&1
python3 tools/trace_summary.py $P/one_stream/*/*_kernel_trace.csv --steps 6 --csv $S/train_step_kernels.csv > $S/train_step_kernels.txt 2>&1
python3 tools/trace_summary.py $P/default/*/*_kernel_trace.csv --steps 6 --csv $S/train_step_kernels_shipped_streams.csv > $S/train_step_kernels_shipped_streams.txt 2>&1
cp $P/fwd/*/*_kernel_stats.csv $S/bench_fwd_kernel_stats.csv; cp $P/default/*/*_kernel_stats.csv $S/bench_default_kernel_stats.csv; cp $P/one_stream/*/*_kernel_stats.csv $S/bench_one_stream_kernel_stats.csv
rm -f $P/pmc_?/*/*_counter_collection.csv $P/pmc_?/*/*_kernel_trace.csv $P/fwd/*/*_kernel_trace.csv
du -sh gpurun_out; echo done
