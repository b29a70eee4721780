cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3d13
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_training.py tests/test_gpu_attention.py -x -q -k "bf16_storage or every_parameter or tau" > gpurun_out/r3d13/tests.txt 2>&1; echo tests rc=$?
python bench.py --workload dense2m --storage bf16 --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/r3d13/dense_bf16.json 2> gpurun_out/r3d13/dense_bf16.err; echo dense rc=$?
python bench.py --storage bf16 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r3d13/default_bf16.json 2> gpurun_out/r3d13/default_bf16.err; echo default rc=$?
# what DDP costs on one GPU: one-rank RCCL rehearsal vs plain, and the deferred join's share
SEG3D_BENCH_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3d13/ddp1.json 2> gpurun_out/r3d13/ddp1.err; echo ddp rc=$?
SEG3D_WGRAD_DEFER=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3d13/nodefer.json 2> gpurun_out/r3d13/nodefer.err; echo nodefer rc=$?
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3d13/plain.json 2> gpurun_out/r3d13/plain.err; echo plain rc=$?
