# The bench lines of DESIGN.md section 6, one file per workload under gpurun_out/r3f/ (run on the GPU box:
#   gpurun --timeout 1150 -- 'bash tools/final_benches.sh').  Every line goes to a file: a silent run is taken for hung.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3f
python bench.py > gpurun_out/r3f/default.json 2> gpurun_out/r3f/default.err; echo default rc=$?
python bench.py --workload dense2m --steps 8 --warmup 3 > gpurun_out/r3f/dense2m.json 2> gpurun_out/r3f/dense2m.err; echo dense rc=$?
python bench.py --workload cylinder --batch 4 --steps 8 --warmup 3 --scenes 2 > gpurun_out/r3f/cylinder.json 2> gpurun_out/r3f/cylinder.err; echo cyl rc=$?
python bench.py --workload multi_sweeps --batch 2 --steps 8 --warmup 3 --scenes 2 > gpurun_out/r3f/multi.json 2> gpurun_out/r3f/multi.err; echo ms rc=$?
python bench.py --segmentor spnet --steps 10 --warmup 3 > gpurun_out/r3f/spnet.json 2> gpurun_out/r3f/spnet.err; echo spnet rc=$?
SEG3D_CONV_PRECISION=fp32 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r3f/fp32.json 2> gpurun_out/r3f/fp32.err; echo fp32 rc=$?
python bench.py --workload dense2m --storage bf16 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r3f/dense2m_bf16.json 2> gpurun_out/r3f/dense2m_bf16.err; echo dense bf16 rc=$?
python bench.py --storage bf16 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r3f/default_bf16.json 2> gpurun_out/r3f/default_bf16.err; echo default bf16 rc=$?
python bench.py --mode fwd --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3f/fwd.json 2> gpurun_out/r3f/fwd.err; echo fwd rc=$?
SEG3D_WGRAD_DEFER=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3f/nodefer.json 2> gpurun_out/r3f/nodefer.err; echo nodefer rc=$?
SEG3D_BENCH_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3f/ddp1.json 2> gpurun_out/r3f/ddp1.err; echo ddp1 rc=$?
python tools/attn_bench.py --bwd > gpurun_out/r3f/attn.txt 2>&1; python tools/attn_bench.py --bwd --drop 0.1 >> gpurun_out/r3f/attn.txt 2>&1; echo attn rc=$?
