cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3d12
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_training.py -x -q -s -k "bf16_storage or hip_graph or every_parameter" > gpurun_out/r3d12/tests.txt 2>&1; echo tests rc=$?
python bench.py --workload dense2m --storage bf16 --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/r3d12/dense_bf16.json 2> gpurun_out/r3d12/dense_bf16.err; echo dense rc=$?
python bench.py --storage bf16 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r3d12/default_bf16.json 2> gpurun_out/r3d12/default_bf16.err; echo default rc=$?
