cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3d14
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_training.py tests/test_gpu_attention.py -x -q -k "bf16_storage or class_context or tau or spnet or rehearsal" > gpurun_out/r3d14/tests.txt 2>&1; echo tests rc=$?
SEG3D_BENCH_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3d14/ddp1.json 2> gpurun_out/r3d14/ddp1.err; echo ddp rc=$?
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3d14/plain.json 2> gpurun_out/r3d14/plain.err; echo plain rc=$?
python bench.py --segmentor spnet --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3d14/spnet.json 2> gpurun_out/r3d14/spnet.err; echo spnet rc=$?
