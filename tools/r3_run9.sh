cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3d9
timeout -k 10 900 python -m pytest tests/test_gpu_integration.py tests/test_gpu_parity.py -x -q -k "integration or config1 or registered or voxel_generator" > gpurun_out/r3d9/tests_new.txt 2>&1; echo new tests rc=$?
timeout -k 10 900 python -m pytest tests/test_gpu_training.py -q -s -k "every_parameter" > gpurun_out/r3d9/tests_grad.txt 2>&1; echo grad rc=$?
timeout -k 10 1100 python -m pytest tests/test_gpu_training.py -x -q -k "full_size_configs" > gpurun_out/r3d9/tests_full.txt 2>&1; echo full rc=$?
