# The bench lines of DESIGN.md section 6, one file per workload under gpurun_out/r5f/ (run on the GPU box in two calls:
#   gpurun --timeout 1150 -- 'bash tools/final_benches.sh a'   and   ... 'bash tools/final_benches.sh b').
# Every line goes to a file: a silent run is taken for hung.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5f; mkdir -p $O
if [ "$1" = "a" ]; then
python bench.py > $O/default.json 2> $O/default.err; echo default rc=$?
python bench.py --mode fwd --steps 20 --warmup 5 --no-cpu-baseline > $O/fwd.json 2> $O/fwd.err; echo fwd rc=$?
SEG3D_WGRAD_CENTER_FIRST=0 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-fp32-exact > $O/no_centre_first.json 2> $O/no_centre_first.err; echo no_centre_first rc=$?
python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-fp32-exact > $O/default_b.json 2> $O/default_b.err; echo default_b rc=$?
SEG3D_WGRAD_DEFER=0 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-fp32-exact > $O/nodefer.json 2> $O/nodefer.err; echo nodefer rc=$?
SEG3D_POINT_MLP=fp32 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-fp32-exact > $O/default_f32mlp.json 2> $O/default_f32mlp.err; echo f32mlp rc=$?
SEG3D_BENCH_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-fp32-exact > $O/ddp1.json 2> $O/ddp1.err; echo ddp1 rc=$?
SEG3D_DDP_OVERLAP=0 SEG3D_BENCH_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29519 bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-fp32-exact > $O/ddp1_nooverlap.json 2> $O/ddp1_nooverlap.err; echo ddp1_nooverlap rc=$?
SEG3D_DDP=torch SEG3D_BENCH_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29518 bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-fp32-exact > $O/ddp1_torch.json 2> $O/ddp1_torch.err; echo ddp1_torch rc=$?
python bench.py --segmentor spnet --steps 10 --warmup 3 > $O/spnet.json 2> $O/spnet.err; echo spnet rc=$?
python tools/attn_bench.py --bwd > $O/attn.txt 2>&1; python tools/attn_bench.py --bwd --drop 0.1 >> $O/attn.txt 2>&1; echo attn rc=$?
python tools/conv_bench.py > $O/conv_layers.txt 2>&1; echo conv rc=$?
else
python bench.py --workload dense2m --steps 8 --warmup 3 > $O/dense2m.json 2> $O/dense2m.err; echo dense rc=$?
python bench.py --workload dense2m --storage bf16 --steps 6 --warmup 2 --no-cpu-baseline > $O/dense2m_bf16.json 2> $O/dense2m_bf16.err; echo dense bf16 rc=$?
python bench.py --workload cylinder --batch 4 --steps 8 --warmup 3 --scenes 2 > $O/cylinder.json 2> $O/cylinder.err; echo cyl rc=$?
python bench.py --workload multi_sweeps --batch 2 --steps 8 --warmup 3 --scenes 2 > $O/multi.json 2> $O/multi.err; echo ms rc=$?
python bench.py --workload multi_sweeps --sweeps 5 --batch 2 --steps 6 --warmup 2 --scenes 2 > $O/multi5.json 2> $O/multi5.err; echo ms5 rc=$?
python bench.py --storage bf16 --steps 6 --warmup 2 --no-cpu-baseline --no-fp32-exact > $O/default_bf16.json 2> $O/default_bf16.err; echo default bf16 rc=$?
fi
