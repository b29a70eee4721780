cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P=gpurun_out/r4t; mkdir -p $P
rocprofv3 --kernel-trace --output-format csv -d $P/default -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-fp32-exact > $P/default.json 2> $P/default.err
echo rc=$?
T=$(ls $P/default/*/*_kernel_trace.csv)
python3 tools/layer_chain.py $T 'attn_fused_bwd<24, 1>' -3 > $P/chain_bwd24.txt 2>&1
python3 tools/layer_chain.py $T 'attn_fused_fwd<24, true>' -3 > $P/chain_fwd24.txt 2>&1
python3 tools/layer_chain.py $T 'attn_fused_bwd<12, 1>' -2 > $P/chain_bwd12.txt 2>&1
SEG3D_WGRAD_STREAM=0 SEG3D_AUX_OVERLAP=0 rocprofv3 --kernel-trace --output-format csv -d $P/one -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-fp32-exact --no-pipeline > $P/one.json 2> $P/one.err
T1=$(ls $P/one/*/*_kernel_trace.csv)
python3 tools/layer_chain.py $T1 'attn_fused_bwd<24, 1>' -3 > $P/chain1_bwd24.txt 2>&1
python3 tools/layer_chain.py $T1 'attn_fused_fwd<24, true>' -3 > $P/chain1_fwd24.txt 2>&1
python3 tools/trace_summary.py $T1 --steps 6 --csv $P/train_step_kernels_one.csv > $P/train_step_kernels_one.txt 2>&1
rm -f $P/*/*/*_kernel_trace.csv
